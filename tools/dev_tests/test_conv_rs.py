"""GPU: the register-stationary-weights conv kernel (csrc/conv_rs.hip, v_mfma_f32_16x16x32_f16) against the
LDS-weights kernel (csrc/conv_mfma.hip) and the CPU oracle, layer shape by layer shape.

Both kernels read and write the same fp16 "planes" tensors and accumulate in fp32, so their results differ only by
the order of the fp32 additions (then one fp16 rounding per layer).  Dev library only since round 5: ``SS4K_RS_MASK=63`` +
``SS4K_CONV5_MODE=1`` (read when a model is built) route every layer shape the kernel is built for to it; without them no layer takes it.
"""
import os

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from tests.helpers import psnr, smooth_u8

pytestmark = pytest.mark.gpu


def _build(ctx, desc, flat, no_rs):
    if not no_rs:
        os.environ.update(SS4K_RS_MASK="63", SS4K_CONV5_MODE="1")
    try:
        return _capi.Model(ctx, desc, flat)
    finally:
        os.environ.pop("SS4K_RS_MASK", None); os.environ.pop("SS4K_CONV5_MODE", None)


@pytest.mark.parametrize("scale,shape", [(2, (1, 3, 64, 96)), (2, (3, 3, 86, 150)), (4, (2, 3, 37, 70)), (1, (1, 3, 128, 256))])
def test_rrdbnet_rs_vs_lds_kernel_and_oracle(ctx, scale, shape):
    """One RRDB block exercises every RS layer shape: conv1-4 (32 couts, 64..160 cin), conv5 (192 -> 64 with
    both residual forms), conv_body / conv_up1 / conv_up2 / conv_hr (64 -> 64, nearest-x2 input addressing);
    odd sizes put ragged tiles and partial pixel blocks on every edge; 3 frames exercise the tile walk."""
    tab = W.rrdbnet_table(61 + scale, scale=scale, num_block=1)
    flat = W.flatten(tab, W.rrdbnet_keys(1))
    desc = _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=1)
    x = torch.from_numpy(smooth_u8(7 + scale, (shape[0], shape[2], shape[3], 3))).permute(0, 3, 1, 2).float().div(255.0)
    y_rs = _build(ctx, desc, flat, no_rs=False)(x.cuda()).cpu()
    y_lds = _build(ctx, desc, flat, no_rs=True)(x.cuda()).cpu()
    with torch.no_grad():
        want = onets.rrdbnet(x, tab, scale, 1)
    peak = float(want.abs().max())
    d = float((y_rs - y_lds).abs().max())
    p_rs, p_lds = psnr(y_rs, want, peak=peak), psnr(y_lds, want, peak=peak)
    print(f"x{scale} {shape}: RS vs LDS kernel max |diff| {d:.3g} (peak {peak:.3g}); PSNR vs oracle RS {p_rs:.1f} dB, LDS {p_lds:.1f} dB")
    assert torch.isfinite(y_rs).all()
    assert d <= 4e-3 * peak            # a few fp16 ulps at the output's magnitude
    assert p_rs > 55.0 and abs(p_rs - p_lds) < 3.0


def test_srvgg_rs_prelu_vs_lds_kernel_and_oracle(ctx):
    """SRVGG body: 64 -> 64 with per-channel PReLU slopes (the <2,8,2,PR> build)."""
    tab = W.srvgg_table(5, num_feat=64, num_conv=4, upscale=2)
    flat = W.flatten(tab, W.srvgg_keys(4))
    desc = _capi.make_desc(_capi.SRVGG, _capi.F16, scale=2, num_feat=64, num_block=4)
    x = torch.rand(2, 3, 53, 77)
    y_rs = _build(ctx, desc, flat, no_rs=False)(x.cuda()).cpu()
    y_lds = _build(ctx, desc, flat, no_rs=True)(x.cuda()).cpu()
    with torch.no_grad():
        want = onets.srvgg(x, tab, 4, 2)
    peak = float(want.abs().max())
    assert float((y_rs - y_lds).abs().max()) <= 4e-3 * peak
    assert psnr(y_rs, want, peak=peak) > 55.0


def test_rs_kernel_is_deterministic_and_batch_invariant(ctx):
    """The tile walk, ring slots and counted waits must never change a pixel: the same frame gives bit-identical
    results alone, inside a batch, and on repeated calls (a landed-too-late DMA would show up here)."""
    tab = W.rrdbnet_table(3, scale=2, num_block=2)
    m = _build(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2), W.flatten(tab, W.rrdbnet_keys(2)), no_rs=False)
    x = torch.rand(4, 3, 360, 640).cuda()
    y = m(x)
    for _ in range(5):
        assert torch.equal(m(x), y)
    assert torch.equal(m(x[2:3]), y[2:3])
    assert torch.equal(m(x[1:4]), y[1:4])


def test_rs_kernel_full_size_repeatability(ctx):
    """Stress of the ring / counted-wait protocol at the headline size: the 23-block network on 4 frames of 720p
    (351 launches, ~7 tiles per workgroup, tile-boundary stores in flight) must give bit-identical frames on
    every one of 12 runs."""
    tab = W.rrdbnet_table(0, scale=2)
    m = _build(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(tab, W.rrdbnet_keys(23)), no_rs=False)
    x = torch.from_numpy(smooth_u8(31, (4, 720, 1280, 3))).permute(0, 3, 1, 2).float().div(255.0).cuda()
    ref = m(x)
    assert torch.isfinite(ref).all()
    for _ in range(11):
        assert torch.equal(m(x), ref)
