"""GPU: conv_dense.hip's single-layer build (conv3x3_wide_kernel: 64 couts per workgroup on the fused kernel's machinery) against
conv_mfma.hip's <__half,2,4,4> build it replaces for fp16 layers with a plain epilogue.  Same tile, same packed fragments, same MFMA
order per output, the same epilogue expressions: results must be BIT-IDENTICAL - through every epilogue form the networks use
(LeakyReLU / PReLU / ReLU6 / none, alpha, one and two residuals written in place, several cout groups, concat inputs, nearest-x2
upsampled input, odd K-chunk counts)."""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W

pytestmark = pytest.mark.gpu
WIDE = _capi.MODEL_NO_W16   # the wide kernel is the route of these layers when conv_w16.hip (the default since round 4, not bit-identical) is off;
                            # set on BOTH sides: it also keeps conv_last off conv_w16n.hip
NO_WIDE, NO_DENSE, ONE, TWO = _capi.MODEL_NO_WIDE, _capi.MODEL_NO_DENSE, _capi.MODEL_ONE_CHAIN, _capi.MODEL_TWO_CHAINS
DIRECT_UPS = _capi.MODEL_NO_UPS_PRESUM   # the pre-summed up-sampling convs are the one route that is not bit-identical: pinned off here


@pytest.mark.parametrize("nf,shape,up", [(64, (2, 3, 72, 130), 4), (64, (1, 3, 33, 47), 2), (128, (1, 3, 40, 64), 2)])
def test_wide_bit_identical_srvgg(ctx, nf, shape, up):
    """SRVGG body: PReLU slopes per channel (select form); 128 features = two cout groups per layer."""
    t = W.dni_blend(W.srvgg_table(3, num_feat=nf, num_conv=4, upscale=up), W.srvgg_table(4, num_feat=nf, num_conv=4, upscale=up), 0.5)
    flat = W.flatten(t, W.srvgg_keys(4))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(nf + shape[3])).cuda()
    outs = [_capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=up, num_feat=nf, num_block=4, flags=fl), flat)(x).clone()
            for fl in (NO_WIDE | WIDE, WIDE)]
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("chns,shape", [((32, 64, 128), (2, 4, 64, 96)), ((64, 128, 256), (1, 4, 48, 80))])
def test_wide_bit_identical_bsvd(ctx, chns, shape):
    """BSVD's half- and quarter-resolution layers: ReLU6, inputs that start at a later plane (the F = 1 BiBufferConv skips its dead
    planes), 128 / 256 couts = several cout groups."""
    t = W.bsvd_table(5, chns=chns)
    flat = W.flatten(t, W.bsvd_keys(chns=chns))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2])).cuda()
    outs = [_capi.Model(ctx, _capi.make_desc(_capi.BSVD, _capi.F16, scale=1, bsvd_chns=chns, flags=fl), flat)(x).clone() for fl in (NO_WIDE | WIDE, WIDE)]
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("scale,shape", [(2, (2, 3, 96, 136)), (4, (1, 3, 45, 67)), (2, (1, 3, 62, 30))])
def test_ups_presum_vs_direct_and_oracle(ctx, scale, shape):
    """conv_up1 / conv_up2 in the pre-summed form (two taps that read the same low-resolution row share one MFMA, their weight
    fragments added in fp16) against the direct form and against the CPU oracle: the two forms agree far inside the fp16 path's own
    error, and the pre-summed one is as close to the oracle as the direct one (odd / ragged output sizes, image borders: zero padding
    of the UP-SAMPLED image, which the pre-summed rows must reproduce)."""
    from oracle import nets as onets
    from tests.helpers import psnr
    t = W.rrdbnet_table(29, scale=scale, num_block=1)
    flat = W.flatten(t, W.rrdbnet_keys(1))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[3]))
    with torch.no_grad():
        want = onets.rrdbnet(x, t, scale, 1)
    direct = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=1, flags=DIRECT_UPS), flat)(x.cuda()).cpu()
    presum = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=1), flat)(x.cuda()).cpu()
    peak = float(want.abs().max())
    p_forms, p_direct, p_presum = psnr(presum / peak, direct / peak), psnr(direct / peak, want / peak), psnr(presum / peak, want / peak)
    print(f"x{scale} {shape}: presum vs direct {p_forms:.1f} dB; vs oracle: direct {p_direct:.1f} dB, presum {p_presum:.1f} dB")
    assert not torch.equal(presum, direct)           # the flag does select another form
    assert p_forms > 66.0 and p_presum > p_direct - 1.0


@pytest.mark.parametrize("shape", [(1, 3, 144, 208), (2, 3, 66, 94)])
def test_conv5_matrix_core_residual_vs_memory_residual_and_oracle(ctx, shape):
    """conv5 of every RDB on the wide kernel carries its residual through the matrix core ((conv + x / alpha) * alpha, one more MFMA per
    accumulator with a (1 / alpha) I fragment); conv_mfma.hip's <2,4,4> build (SS4K_MODEL_NO_WIDE - what a job with planes beyond 4 GB gets)
    reads it from memory in the epilogue: another order of fp32 additions, the same accuracy against the oracle; every RDB (one residual, and
    two with the block's input written in place).  (Default routing: conv_w16.hip's build of the same form - tests/test_gpu_w16.py.  The
    trunk / tail layers of an RRDBNet on the two kernels, bit for bit: tools/dev_tests/test_wide_rrdbnet.py, which can switch the
    matrix-core residual off.)"""
    from oracle import nets as onets
    from tests.helpers import psnr
    t = W.rrdbnet_table(33, scale=2, num_block=3)
    flat = W.flatten(t, W.rrdbnet_keys(3))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[3]))
    with torch.no_grad():
        want = onets.rrdbnet(x, t, 2, 3)
    rl = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=3, flags=WIDE | DIRECT_UPS), flat)(x.cuda()).cpu()
    mem = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=3, flags=WIDE | NO_WIDE | DIRECT_UPS), flat)(x.cuda()).cpu()
    peak = float(want.abs().max())
    p_routes, p_rl, p_mem = psnr(rl / peak, mem / peak), psnr(rl / peak, want / peak), psnr(mem / peak, want / peak)
    print(f"{shape}: routes {p_routes:.1f} dB apart; vs oracle: matrix-core residual {p_rl:.2f} dB, residual from memory {p_mem:.2f} dB")
    assert not torch.equal(rl, mem) and p_routes > 70.0 and abs(p_rl - p_mem) < 0.5
