"""GPU: ``ss4k_op_cv_area_resize_u8`` (the image server's pre / post scale, ``image_pipeline.py:272-273,347-348``) against ``oracle/cv_area.py``, byte for
byte.  Parity UNPINNED: the oracle restates OpenCV's published algorithm, cv2 itself is not in the image."""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from oracle import cv_area as A

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,fx,fy", [((1, 37, 53, 3), 0.8, None), ((2, 64, 96, 3), 0.85, None), ((1, 125, 77, 3), 0.66, None), ((3, 40, 50, 1), 0.8, 0.66),
                                         ((1, 33, 129, 4), 0.85, 0.8), ((1, 3, 3, 3), 0.66, None), ((1, 720, 1280, 3), 0.8, None), ((1, 1440, 2560, 3), 0.85, None)])
def test_matches_the_oracle_bit_for_bit(ctx, shape, fx, fy):
    x = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    got = ctx.cv_area_resize(torch.from_numpy(x).cuda(), fx, fy).cpu().numpy()
    want = np.stack([A.resize_area(x[i], fx, fy) for i in range(shape[0])])
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {want.size} bytes differ, max {int(np.abs(got.astype(int) - want.astype(int)).max())}"


def test_maximum_request_size_and_the_post_scale_of_its_result(ctx):
    """The image server's largest request (4096 x 2048, image_pipeline.py:264) pre-scaled by 0.8, and a 4K result post-scaled by 0.85."""
    x = np.random.default_rng(7).integers(0, 256, (1, 2048, 4096, 3), dtype=np.uint8)
    d = torch.from_numpy(x).cuda()
    pre = ctx.cv_area_resize(d, 0.8)
    assert tuple(pre.shape) == (1, 1638, 3277, 3)
    assert np.array_equal(pre.cpu().numpy()[0], A.resize_area(x[0], 0.8))
    post = ctx.cv_area_resize(pre, 0.85)
    assert np.array_equal(post.cpu().numpy()[0], A.resize_area(pre.cpu().numpy()[0], 0.85))


def test_bad_arguments_and_the_table_cache(ctx):
    d = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device="cuda")
    for f in (0.5, 1.0, 1.25):
        with pytest.raises(_capi.Ss4kError):
            ctx.cv_area_resize(d, f)
    with pytest.raises(_capi.Ss4kError, match="channels"):
        ctx.cv_area_resize(torch.zeros((1, 16, 16, 5), dtype=torch.uint8, device="cuda"), 0.8)
    out = torch.empty(10, dtype=torch.uint8, device="cuda")
    assert _capi.lib().ss4k_op_cv_area_resize_u8(ctx._h, d.data_ptr(), out.data_ptr(), out.numel(), 1, 16, 16, 3, 0.8, 0.8, None) == -22   # too small
    # more shapes than the per-context table cache holds (64): results stay right across the flush, from two streams
    rng = np.random.default_rng(3)
    side = torch.cuda.Stream()
    for k in range(70):
        h, w = 20 + k, 31 + (k % 7)
        x = rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)
        dx = torch.from_numpy(x).cuda()
        if k % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = ctx.cv_area_resize(dx, 0.8)
            side.synchronize()
        else:
            got = ctx.cv_area_resize(dx, 0.8)
        assert np.array_equal(got.cpu().numpy()[0], A.resize_area(x[0], 0.8)), k
