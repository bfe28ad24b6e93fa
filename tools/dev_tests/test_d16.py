"""GPU, dev library (SS4K_LIB=libss4k_hip_dev.so, SS4K_D16=1; run by tests/test_gpu_dev_kernels.py): conv_d16.hip - the fused dense-block
layer pairs on v_mfma_f32_16x16x32_f16, 14 x 32 tiles, the ring of x_k as three gathered 16-pixel groups per wave - against the four
launches it replaces and against the CPU oracle.

Not bit-identical to either 32x32x16 route (an MFMA adds 32 products of two taps where the other adds 16 of one); the results must sit
within fp32 accumulation noise ahead of each layer's fp16 rounding - far inside the fp16 path's own distance to the oracle.  What the
comparison catches at that tolerance: a wrong halo (x_k evaluated on padded input instead of zeros), a stale LDS image, a ring pixel
mapped to the wrong place, a store by a non-owner lane, a missed tile or row of the 14-row tiling - each changes whole pixels."""
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from tests.helpers import psnr

pytestmark = pytest.mark.gpu
NO_DENSE, ONE, TWO, NO_W16 = _capi.MODEL_NO_DENSE, _capi.MODEL_ONE_CHAIN, _capi.MODEL_TWO_CHAINS, _capi.MODEL_NO_W16


def _model(ctx, flat, scale, nb, flags, **kw):
    return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=nb, flags=flags, **kw), flat)


# interior grids (h / r, w / r) with one to many tiles of 14 x 32, ragged right / bottom edges, heights around multiples of 14, widths
# around multiples of 32, one- to four-frame jobs (one and two launch chains)
@pytest.mark.parametrize("scale,shape,lanes", [(2, (1, 3, 144, 208), ONE), (2, (2, 3, 92, 200), TWO), (2, (3, 3, 80, 72), ONE),
                                               (4, (1, 3, 37, 70), ONE), (1, (1, 3, 128, 256), ONE), (2, (4, 3, 56, 128), TWO),
                                               (2, (1, 3, 28, 64), ONE), (2, (1, 3, 30, 66), ONE), (2, (2, 3, 26, 62), TWO),
                                               (4, (2, 3, 9, 33), ONE), (4, (1, 3, 14, 32), ONE), (4, (1, 3, 15, 33), ONE),
                                               (2, (1, 3, 360, 500), ONE), (4, (1, 3, 1, 1), ONE), (4, (2, 3, 2, 95), TWO)])
def test_d16_vs_four_launches_and_oracle(ctx, scale, shape, lanes):
    t = W.rrdbnet_table(11, scale=scale, num_block=2)
    flat = W.flatten(t, W.rrdbnet_keys(2))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] * 1000 + shape[3]))
    with torch.no_grad():
        want = onets.rrdbnet(x, t, scale, 2)
    four = _model(ctx, flat, scale, 2, NO_DENSE | lanes)(x.cuda()).cpu()
    m = _model(ctx, flat, scale, 2, lanes)
    got = m(x.cuda()).cpu()
    peak = float(want.abs().max())
    p_forms, p_four, p_got = psnr(got / peak, four / peak), psnr(four / peak, want / peak), psnr(got / peak, want / peak)
    print(f"x{scale} {shape}: d16 vs four launches {p_forms:.1f} dB, max |d| {float((got - four).abs().max()) / peak:.2e} of peak; vs oracle {p_four:.1f} / {p_got:.1f} dB")
    assert torch.isfinite(got).all() and (shape[2] * shape[3] < 64 or not torch.equal(got, four))   # (a one-pixel image sums too little to differ)
    assert p_forms > 66.0 and float((got - four).abs().max()) < 6e-3 * peak and p_got > p_four - 0.5
    for _ in range(2):   # repeated calls: the LDS images / buffers of one launch must not leak into the next
        assert torch.equal(m(x.cuda()).cpu(), got)


def test_d16_720p_23_blocks_vs_four_launches_and_job_size(ctx):
    """The headline network at full size: 4 frames of 720p through 23 blocks, two launch chains; a one-frame job reproduces the frame."""
    flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
    x = torch.rand(4, 3, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
    four = _model(ctx, flat, 2, 23, NO_DENSE)(x).clone()
    m = _model(ctx, flat, 2, 23, 0)
    got = m(x).clone()
    peak = float(four.abs().max())
    p = psnr(got / peak, four / peak)
    print(f"23 blocks, 4 x 720p: d16 vs four launches {p:.1f} dB")
    assert torch.isfinite(got).all() and p > 55.0
    for i in range(3):
        assert torch.equal(m(x), got), f"run {i}: output changed"
    assert torch.equal(m(x[:1].contiguous()), got[:1])


@pytest.mark.parametrize("nf,g", [(32, 32), (96, 32)])
def test_d16_other_widths(ctx, nf, g):
    """Trunk widths other than RealESRGAN's 64: conv_k has 2 / 6 (and conv3 4 / 8) K-chunks = 1 / 3 (2 / 4) chunk pairs."""
    t = W.rrdbnet_table(21, scale=2, num_feat=nf, num_block=1, num_grow_ch=g)
    flat = W.flatten(t, W.rrdbnet_keys(1))
    x = torch.rand(2, 3, 72, 136, generator=torch.Generator().manual_seed(nf + g))
    outs = [_model(ctx, flat, 2, 1, fl | ONE, num_feat=nf, num_grow_ch=g)(x.cuda()).cpu() for fl in (NO_DENSE, 0)]
    peak = float(outs[0].abs().max())
    assert torch.isfinite(outs[1]).all() and not torch.equal(outs[0], outs[1]) and psnr(outs[1] / peak, outs[0] / peak) > 66.0
