"""GPU: the fused layer-pair kernel (csrc/conv_pair.hip) - BSVD's full-resolution inc and outc blocks as one row-marching launch
each, the 32-channel tensor between the two layers held in LDS - against the two launches it replaces (SS4K_MODEL_NO_PAIR).
Same packed weights, same MFMA order, same fp16 rounding of the inter tensor: the frames must be BIT-IDENTICAL."""
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.upscale import model as factory
from oracle import nets as onets
from tests.helpers import psnr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 4, 32, 48), (2, 4, 64, 124), (3, 4, 36, 128), (4, 4, 180, 320), (1, 4, 8, 64), (2, 4, 100, 252),
                                   (1, 4, 720, 1280)])
def test_pair_bit_identical_to_two_launches(ctx, shape):
    """Strips of 62 columns: widths below, at and across strip borders (48, 124 = 2 x 62, 128, 252, 320, 1280); heights that split into
    one and several bands; odd and even frame counts (frame lanes split an even job over two streams)."""
    tab = W.bsvd_table(seed=21)
    fused = factory.build_denoise_model(ctx, weights=tab, dtype="f16")
    plain = factory.build_denoise_model(ctx, weights=tab, dtype="f16", flags=_capi.MODEL_NO_PAIR)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] + shape[3]))
    x[:, 3] = 0.1 * x[:, 3]   # the noise map
    xg = x.cuda()
    want = plain(xg).clone()
    assert torch.isfinite(want).all()
    for _ in range(3):   # the lane decision is measured over the first calls: both schedules are covered
        got = fused(xg)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"{shape}: fused pair differs, max |d| {float((got - want).abs().max()):.3g}"


def test_pair_bsvd64_falls_back(ctx):
    """bsvd-64 (64-channel full-resolution layers) does not fit the fused kernel: the executor takes the two launches, results as before."""
    kw = factory.BSVD_VARIANTS["bsvd-64"]
    tab = W.bsvd_table(seed=33, **kw)
    x = torch.rand(1, 4, 32, 48, generator=torch.Generator().manual_seed(1))
    a = factory.build_denoise_model(ctx, weights=tab, dtype="f16", variant="bsvd-64")(x.cuda()).clone()
    b = factory.build_denoise_model(ctx, weights=tab, dtype="f16", variant="bsvd-64", flags=_capi.MODEL_NO_PAIR)(x.cuda())
    assert torch.equal(a, b)
    with torch.no_grad():
        want = onets.bsvd_f1(x[:, None], tab)[:, 0]
    assert psnr(a.float().cpu(), want) > 45.0
