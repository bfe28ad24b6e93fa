"""CPU: the restatement of cv2's INTER_AREA shrink (``oracle/cv_area.py``; parity UNPINNED - cv2 is not in the image, so these are the properties
the published algorithm must have, two hand-computed cases, and the library's host-side shape function)."""
import ctypes as C

import numpy as np
import pytest

from oracle import cv_area as A


@pytest.mark.parametrize("f", [0.8, 0.85, 0.66])
def test_constant_image_stays_constant_and_shape_is_cvround(f):
    for h, w in ((37, 53), (64, 64), (125, 3), (1, 17)):
        img = np.full((h, w, 3), 201, np.uint8)
        out = A.resize_area(img, f)
        assert out.shape == (int(np.rint(h * f)), int(np.rint(w * f)), 3) and (out == 201).all()


def test_shape_rounds_half_to_even():
    assert A.out_size(25, 0.1) == 2 and A.out_size(35, 0.1) == 4 and A.out_size(15, 0.1) == 2   # 2.5 -> 2, 3.5 -> 4, 1.5 -> 2


@pytest.mark.parametrize("n,f", [(53, 0.8), (1001, 0.8), (640, 0.85), (77, 0.66), (4096, 0.8), (2048, 0.85), (10, 0.66)])
def test_table_covers_every_source_pixel_once_in_total(n, f):
    d = A.out_size(n, f)
    tab = A.area_tab(n, d, 1.0 / f)
    per_cell, per_src = np.zeros(d), np.zeros(n)
    last = (-1, -1)
    for di, si, a in tab:
        assert 0 <= si < n and 0 <= di < d and a > 0
        assert (di, si) > last   # table order: cells ascending, sources ascending inside a cell
        last = (di, si)
        per_cell[di] += float(a)
        per_src[si] += float(a) * min(1.0 / f, n - di / f)
    assert np.allclose(per_cell, 1.0, atol=1e-6)           # a cell's weights are a partition of one
    covered = min(n, d / f)                                  # the cells cover [0, d / f) of the source, cut at its end
    assert abs(per_src.sum() - covered) < 1e-3 * n
    assert np.all(per_src[: int(covered) - 1] > 1.0 - 1e-3)   # every whole source pixel inside the covered span is used exactly once in total


def test_hand_computed_1d_cases():
    # f = 0.8 -> cells of 1.25 source pixels: [0, 1.25) = 0.8 p0 + 0.2 p1; [1.25, 2.5) = 0.6 p1 + 0.4 p2; [2.5, 3.75) = 0.4 p2 + 0.6 p3; [3.75, 5) = 0.2 p3 + 0.8 p4
    row = np.array([10, 20, 40, 80, 160], np.uint8).reshape(1, 5, 1)
    img = np.repeat(row, 5, axis=0)
    out = A.resize_area(img, 0.8)
    assert out.shape == (4, 4, 1)
    assert out[0, :, 0].tolist() == [12, 28, 64, 144]
    # f = 0.66 -> scale 1.515...: 3 source pixels -> cvRound(1.98) = 2 cells; the second cell is cut at the image's end: cellWidth = 3 - 1.5152 = 1.4848
    row = np.array([0, 100, 200], np.uint8).reshape(1, 3, 1)
    out = A.resize_area(np.repeat(row, 3, axis=0), 0.66)
    s = 1 / 0.66
    want0 = (1.0 * 0 + (s - 1) * 100) / s
    want1 = ((2 - s) * 100 + 1.0 * 200) / (3 - s)
    assert out.shape == (2, 2, 1) and out[0, :, 0].tolist() == [int(np.rint(want0)), int(np.rint(want1))]


def test_out_of_scope_factors_raise():
    img = np.zeros((8, 8, 3), np.uint8)
    for f in (0.5, 0.25, 1.0, 1.5, 0.0):
        with pytest.raises(ValueError):
            A.resize_area(img, f)


def test_library_shape_function_matches_the_oracle_and_rejects_the_same_factors():
    import sharkshark4k_amd  # noqa: F401
    from sharkshark4k_amd import _capi
    L = _capi.lib()
    oh, ow = C.c_int(), C.c_int()
    for h, w, f in ((720, 1280, 0.8), (1440, 2560, 0.85), (37, 53, 0.66), (2048, 4096, 0.8), (25, 35, 0.1 + 1e-9)):
        assert L.ss4k_op_cv_area_shape(h, w, f, f, C.byref(oh), C.byref(ow)) == 0
        assert (oh.value, ow.value) == (A.out_size(h, f), A.out_size(w, f))
    for f in (0.5, 1.0, 2.0, 0.0, -0.3):
        assert L.ss4k_op_cv_area_shape(64, 64, f, 0.8, C.byref(oh), C.byref(ow)) == -22
    assert b"integer" in L.ss4k_last_error() or b"shrink" in L.ss4k_last_error()
