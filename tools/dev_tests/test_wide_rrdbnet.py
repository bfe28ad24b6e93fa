"""GPU, dev library: conv_dense.hip's single-layer build (conv3x3_wide_kernel) against conv_mfma.hip's <__half,2,4,4> build through a whole
RRDBNet, bit for bit - trunk / tail layers (incl. both up-sampling convs in the direct form) and conv5 of every RDB with its residual READ
FROM MEMORY on both sides (SS4K_NO_RL=1, a dev-library switch read when a model is built: the product always carries conv5's residual
through the matrix core on the 64-cout tile, a form <2,4,4> does not have).  The SRVGG / BSVD halves of this comparison need no switch
and run in tests/test_gpu_wide.py."""
import os

import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W

pytestmark = pytest.mark.gpu
WIDE = _capi.MODEL_NO_W16
NO_WIDE, NO_DENSE, ONE, TWO = _capi.MODEL_NO_WIDE, _capi.MODEL_NO_DENSE, _capi.MODEL_ONE_CHAIN, _capi.MODEL_TWO_CHAINS
DIRECT_UPS = _capi.MODEL_NO_UPS_PRESUM


def _model(ctx, flat, scale, flags):
    os.environ["SS4K_NO_RL"] = "1"
    try:
        return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=2, flags=flags), flat)
    finally:
        os.environ.pop("SS4K_NO_RL", None)


@pytest.mark.parametrize("scale,shape,base", [(2, (1, 3, 144, 208), ONE), (2, (2, 3, 92, 200), TWO), (4, (1, 3, 37, 70), ONE),
                                              (1, (1, 3, 128, 256), ONE), (4, (3, 3, 9, 33), ONE), (2, (2, 3, 34, 62), NO_DENSE | TWO)])
def test_wide_bit_identical_rrdbnet(ctx, scale, shape, base):
    t = W.rrdbnet_table(17, scale=scale, num_block=2)
    flat = W.flatten(t, W.rrdbnet_keys(2))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] * 7 + shape[3])).cuda()
    want = _model(ctx, flat, scale, base | DIRECT_UPS | NO_WIDE | WIDE)(x).clone()
    m = _model(ctx, flat, scale, base | DIRECT_UPS | WIDE)
    for _ in range(2):
        got = m(x)
        assert torch.isfinite(got).all() and torch.equal(got, want), f"{shape} flags {base}: max |d| {float((got - want).abs().max()):.3g}"
