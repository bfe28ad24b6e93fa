"""GPU: conv_w16.hip (conv3x3_w16_kernel: the 64-cout single-layer tile on v_mfma_f32_16x16x32_f16, the default route; SS4K_MODEL_NO_W16 selects the other) against
conv_dense.hip's wide kernel (v_mfma_f32_32x32x16_f16) it stands in for, and against the CPU oracle.

The two kernels sum the same products in a different order (an MFMA adds 32 products of two taps where the other adds 16 of one), so
their results are NOT bit-identical; they must agree to within fp32 accumulation noise ahead of the fp16 rounding of each layer's
output - far inside the fp16 path's own distance to the oracle - through every epilogue form the networks use (PReLU per channel,
LeakyReLU, ReLU6, none, alpha, one and two residuals written in place, several cout groups, concat inputs), on ragged sizes
(partly filled tiles on every edge), one to many tiles per workgroup, one and two launch chains."""
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from tests.helpers import psnr

pytestmark = pytest.mark.gpu
NO_W16, ONE, TWO = _capi.MODEL_NO_W16, _capi.MODEL_ONE_CHAIN, _capi.MODEL_TWO_CHAINS


def _close(got, ref, want, what, db_forms, slack=0.5):
    """got (w16) vs ref (wide): db_forms apart at least; and got as close to the oracle as ref is (within `slack` dB)."""
    peak = float(want.abs().max())
    p_forms, p_ref, p_got = psnr(got / peak, ref / peak), psnr(ref / peak, want / peak), psnr(got / peak, want / peak)
    print(f"{what}: w16 vs wide {p_forms:.1f} dB; vs oracle: wide {p_ref:.1f} dB, w16 {p_got:.1f} dB")
    assert torch.isfinite(got).all()
    assert not torch.equal(got, ref), f"{what}: the flag selected no other kernel"
    assert p_forms > db_forms and p_got > p_ref - slack, (what, p_forms, p_ref, p_got)


@pytest.mark.parametrize("nf,shape,up,lanes", [(64, (2, 3, 72, 130), 4, TWO), (64, (1, 3, 33, 47), 2, ONE), (128, (1, 3, 40, 64), 2, ONE),
                                               (64, (3, 3, 16, 32), 4, TWO), (64, (1, 3, 150, 331), 2, ONE), (64, (1, 3, 5, 7), 4, ONE)])
def test_w16_srvgg_vs_wide_and_oracle(ctx, nf, shape, up, lanes):
    """SRVGG body (every 64 -> 64 + PReLU layer; 128 features = two cout groups and eight K-chunks)."""
    t = W.dni_blend(W.srvgg_table(3, num_feat=nf, num_conv=4, upscale=up), W.srvgg_table(4, num_feat=nf, num_conv=4, upscale=up), 0.5)
    flat = W.flatten(t, W.srvgg_keys(4))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(nf + shape[3]))
    with torch.no_grad():
        want = onets.srvgg(x, t, upscale=up, num_conv=4)
    outs = [_capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=up, num_feat=nf, num_block=4, flags=lanes | fl), flat)(x.cuda()).cpu()
            for fl in (NO_W16, 0)]
    _close(outs[1], outs[0], want, f"srvgg {nf} {shape}", 66.0)
    m = _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=up, num_feat=nf, num_block=4, flags=lanes), flat)
    for _ in range(3):
        assert torch.equal(m(x.cuda()).cpu(), outs[1]), "w16: output changed between calls"


@pytest.mark.parametrize("scale,shape,base", [(2, (1, 3, 144, 208), ONE), (2, (2, 3, 92, 200), TWO), (4, (1, 3, 37, 70), ONE), (1, (1, 3, 128, 256), ONE)])
def test_w16_rrdbnet_vs_wide_and_oracle(ctx, scale, shape, base):
    """RRDBNet: trunk conv (+ feat residual read from memory), the high-resolution convs, and conv5 of every RDB (192 -> 64: twelve K-chunks
    from two tensors, x 0.2 + x through the matrix core - the RL builds of the two kernels - and the RRDB's second residual written in place)."""
    t = W.rrdbnet_table(17, scale=scale, num_block=2)
    flat = W.flatten(t, W.rrdbnet_keys(2))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] * 7 + shape[3]))
    with torch.no_grad():
        want = onets.rrdbnet(x, t, scale, 2)
    for extra in (0, _capi.MODEL_NO_DENSE):
        outs = [_capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=2, flags=base | extra | fl), flat)(x.cuda()).cpu()
                for fl in (NO_W16, 0)]
        _close(outs[1], outs[0], want, f"rrdbnet x{scale} {shape} flags {extra}", 60.0)


@pytest.mark.parametrize("chns,shape", [((32, 64, 128), (2, 4, 64, 96)), ((64, 128, 256), (1, 4, 48, 80))])
def test_w16_bsvd_vs_wide_and_oracle(ctx, chns, shape):
    """BSVD's half- and quarter-resolution layers: ReLU6, inputs that start at a later plane, 128 / 256 couts = several cout groups."""
    t = W.bsvd_table(5, chns=chns)
    flat = W.flatten(t, W.bsvd_keys(chns=chns))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2]))
    with torch.no_grad():
        want = onets.bsvd_f1(x[:, None], t)[:, 0]
    outs = [_capi.Model(ctx, _capi.make_desc(_capi.BSVD, _capi.F16, scale=1, bsvd_chns=chns, flags=fl), flat)(x.cuda()).cpu() for fl in (NO_W16, 0)]
    _close(outs[1], outs[0], want, f"bsvd {chns} {shape}", 60.0)


def test_w16_prelu_slopes_above_one_and_below_zero(ctx):
    """The PReLU epilogue has two forms: max(t, t s) when every slope of the layer is <= 1 (host-checked at model build) and the select
    t >= 0 ? t : t s otherwise.  Slopes in [-0.5, 1.7] put layers on both; either must match the oracle as the other route does."""
    import numpy as np
    t = dict(W.srvgg_table(9, num_feat=64, num_conv=4, upscale=2))
    rng = np.random.default_rng(4)
    n_sel = 0
    for k in list(t):
        a = np.asarray(t[k])
        if a.ndim == 1 and k.endswith(".weight"):   # PReLU slopes
            lo, hi = (-0.5, 1.7) if n_sel % 2 == 0 else (-0.5, 1.0)
            t[k] = rng.uniform(lo, hi, a.shape).astype(np.float32)
            n_sel += 1
    assert n_sel >= 4
    flat = W.flatten(t, W.srvgg_keys(4))
    x = torch.rand(2, 3, 40, 70, generator=torch.Generator().manual_seed(2)) - 0.3
    with torch.no_grad():
        want = onets.srvgg(x, t, upscale=2, num_conv=4)
    outs = [_capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=2, num_feat=64, num_block=4, flags=fl), flat)(x.cuda()).cpu()
            for fl in (NO_W16, 0)]
    _close(outs[1], outs[0], want, "srvgg, slopes in [-0.5, 1.7]", 60.0)
