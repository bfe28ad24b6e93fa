"""GPU: the fused dense-block layer pairs (csrc/conv_dense.hip) - (conv1, conv2) and (conv3, conv4) of every RDB as ONE launch
each, the shared input planes streamed once and x1 / x3 handed over in LDS - against the four launches they replace.

The fused kernel runs the same packed fragments in the same (K-chunk, dx, dy) order per output, starts its fp32 accumulators from
the bias, rounds x_k to fp16 exactly as the store does and zero-pads x_k outside the image: results must be BIT-IDENTICAL to
SS4K_MODEL_NO_DENSE (one launch per layer on the LDS-weights kernel).  A wrong halo (x_k evaluated on padded input instead of
zeros), a stale LDS image, a store by a non-owner lane or a missed tile changes the network's output."""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from tests.helpers import psnr

pytestmark = pytest.mark.gpu

NO_DENSE, DENSE, ONE, TWO = _capi.MODEL_NO_DENSE, 0, _capi.MODEL_ONE_CHAIN, _capi.MODEL_TWO_CHAINS   # DENSE: the default (fused pairs)


PIN32 = _capi.MODEL_NO_W16   # this file tests conv_dense.hip (v_mfma_f32_32x32x16_f16: bit-identical to four launches of conv_mfma.hip); the default
                             # fused route since round 4 is conv_d16.hip (16x16x32, another summation order): tests/test_gpu_d16.py


def _model(ctx, flat, scale, nb, flags):
    return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=nb, flags=flags | PIN32), flat)


# shapes: interior grid (h / r, w / r) with 1 .. many tiles of 16 x 30, ragged right / bottom edges, widths just below / at / above
# a multiple of 30, heights that are not multiples of 16, one- to four-frame jobs (lanes on and off)
@pytest.mark.parametrize("scale,shape,lanes", [(2, (1, 3, 144, 208), ONE), (2, (2, 3, 92, 200), TWO), (2, (3, 3, 80, 72), ONE),
                                               (4, (1, 3, 37, 70), ONE), (1, (1, 3, 128, 256), ONE), (2, (4, 3, 64, 120), TWO),
                                               (2, (1, 3, 32, 60), ONE), (2, (1, 3, 34, 62), ONE), (2, (2, 3, 30, 58), TWO),
                                               (4, (2, 3, 9, 33), ONE), (4, (1, 3, 16, 30), ONE), (4, (1, 3, 17, 31), ONE),
                                               (2, (1, 3, 360, 500), ONE), (4, (1, 3, 1, 1), ONE), (4, (2, 3, 2, 95), TWO)])
def test_dense_pair_bit_identical_to_four_launches(ctx, scale, shape, lanes):
    t = W.rrdbnet_table(11, scale=scale, num_block=2)
    flat = W.flatten(t, W.rrdbnet_keys(2))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] * 1000 + shape[3])).cuda()
    want = _model(ctx, flat, scale, 2, NO_DENSE | lanes)(x).clone()
    m = _model(ctx, flat, scale, 2, DENSE | lanes)
    for _ in range(3):   # repeated calls: the LDS images / buffers of one launch must not leak into the next
        got = m(x).clone()
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), f"{shape}: fused pairs differ from four launches, max |d| {float((got - want).abs().max()):.3g}"


def test_dense_pair_720p_23_blocks_bit_identical_and_repeatable(ctx):
    """The headline network at full size: 4 frames of 720p through 23 blocks (138 fused launches per lane), two launch chains."""
    flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
    x = torch.rand(4, 3, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
    PIN = 0   # (conv5 runs on one kernel for every job size: the 1-frame job below must reproduce the 4-frame job's frame)
    want = _model(ctx, flat, 2, 23, NO_DENSE | PIN)(x).clone()
    m = _model(ctx, flat, 2, 23, DENSE | PIN)
    for i in range(4):
        got = m(x)
        assert torch.equal(got, want), f"run {i}: fused pairs differ from four launches"
    one = m(x[:1].contiguous())
    assert torch.equal(one, want[:1])


def test_dense_pair_route_matches_oracle(ctx):
    """Small network against the CPU oracle: the fused route is as close as the four-launch route (it is the same arithmetic)."""
    t = W.rrdbnet_table(3, scale=2, num_block=2)
    flat = W.flatten(t, W.rrdbnet_keys(2))
    x = torch.rand(1, 3, 96, 136, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = onets.rrdbnet(x, t, 2, 2)
    got = _model(ctx, flat, 2, 2, DENSE)(x.cuda()).cpu()
    assert psnr(got, want) > 60.0


@pytest.mark.parametrize("nf,g", [(32, 32), (96, 32), (64, 64)])
def test_dense_pair_other_widths_bit_identical(ctx, nf, g):
    """Trunk widths other than RealESRGAN's 64 / 32: conv_k has 2 / 6 / 4 (and conv3 6 / 10 / 12) K-chunks - the kernel build that
    reads the chunk count from its arguments; a 64-channel growth is not a 32-cout pair and must take four launches by itself."""
    t = W.rrdbnet_table(21, scale=2, num_feat=nf, num_block=1, num_grow_ch=g)
    flat = W.flatten(t, W.rrdbnet_keys(1))
    x = torch.rand(2, 3, 72, 136, generator=torch.Generator().manual_seed(nf + g)).cuda()
    outs = []
    for fl in (NO_DENSE | ONE, DENSE | ONE):
        m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_feat=nf, num_block=1, num_grow_ch=g, flags=fl | PIN32), flat)
        outs.append(m(x).clone())
    assert torch.isfinite(outs[1]).all() and torch.equal(outs[0], outs[1])
