"""CPU oracle: ``cv2.resize(img, None, fx=s, fy=s, interpolation=cv2.INTER_AREA)`` on uint8 HWC images, 0 < s < 1 with 1 / s not an integer.  TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  The image server scales its input down before the upscaler and the result down after it with this call
(``/root/reference/src/sharkshark/image_server/image_pipeline.py:272-273,347-348``; the factors: ``post_scale = 0.66`` by default
``:149-150``, ``pre_scale = 0.8`` / ``post_scale = 0.85`` above one megapixel ``:259-261``).  The arithmetic is OpenCV's (dependency
``opencv-python``, imported as ``cv2`` at ``image_pipeline.py:9``; the reference pins no version and the package is absent from this
image), so this file restates the PUBLISHED algorithm of OpenCV 4.x ``modules/imgproc/src/resize.cpp`` and nothing in the repository can
check it against the real library:

* ``cv::resize``: with ``dsize`` empty, ``dsize = (cvRound(w * fx), cvRound(h * fy))`` (round half to even) and the source step per output
  pixel is ``scale = 1 / fx`` exactly (not ``w / dsize``);
* ``scale`` is not an integer, so the general area path runs: ``computeResizeAreaTab`` turns every output cell ``[d * scale, (d + 1) * scale)``
  into up to three kinds of entries - a partial first source pixel, whole pixels, a partial last pixel - with float32 weights
  ``alpha = covered / cellWidth``, ``cellWidth = min(scale, ssize - d * scale)``, partial pixels only when they cover more than 1e-3;
* ``ResizeArea_Invoker<uchar, float>``: per source row ``buf[dx] += S[sx] * alpha`` over the x entries in table order (float32, multiply then add),
  per output row ``sum = beta * buf`` for the first y entry and ``sum += beta * buf`` for the others, ``dst = saturate_cast<uchar>(sum)`` =
  round half to even, clipped.

Integer ``1 / fx`` (OpenCV's separate fast path, other rounding) and ``fx >= 1`` (INTER_AREA then interpolates linearly) are out of this file's
scope - the reference never asks for them - and raise.
"""
import math

import numpy as np


def out_size(src: int, f: float) -> int:
    """``saturate_cast<int>(src * f)``: round half to even."""
    return int(np.rint(src * f))


def area_tab(ssize: int, dsize: int, scale: float):
    """``computeResizeAreaTab`` -> [(di, si, float32 alpha)] in table order."""
    tab = []
    for d in range(dsize):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, ssize - fs1)
        s1, s2 = math.ceil(fs1), math.floor(fs2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        if s1 - fs1 > 1e-3:
            tab.append((d, s1 - 1, np.float32((s1 - fs1) / cell)))
        for s in range(s1, s2):
            tab.append((d, s, np.float32(1.0 / cell)))
        if fs2 - s2 > 1e-3:
            tab.append((d, s2, np.float32(min(min(fs2 - s2, 1.0), cell) / cell)))
    return tab


def check_scale(f: float) -> float:
    if not (0.0 < f < 1.0):
        raise ValueError(f"INTER_AREA restated for shrinking only (0 < f < 1), got {f}")
    scale = 1.0 / f
    if abs(scale - round(scale)) < np.finfo(np.float64).eps:
        raise ValueError(f"1 / f = {scale} is an integer: OpenCV takes its fast path there (other rounding), not restated")
    return scale


def resize_area(img: np.ndarray, fx: float, fy: float = None) -> np.ndarray:
    """img: uint8 (H, W, C) -> uint8 (cvRound(H * fy), cvRound(W * fx), C)."""
    fy = fx if fy is None else fy
    assert img.dtype == np.uint8 and img.ndim == 3
    sx, sy = check_scale(fx), check_scale(fy)
    h, w, c = img.shape
    oh, ow = out_size(h, fy), out_size(w, fx)
    if oh < 1 or ow < 1:
        raise ValueError("empty output")
    xtab, ytab = area_tab(w, ow, sx), area_tab(h, oh, sy)
    src = img.astype(np.float32)
    # horizontal pass of EVERY source row at once (the invoker does it row by row; the arithmetic per element is the same)
    buf = np.zeros((h, ow, c), np.float32)
    for di, si, a in xtab:
        buf[:, di, :] = buf[:, di, :] + src[:, si, :] * a
    out = np.zeros((oh, ow, c), np.float32)
    first = np.ones(oh, bool)
    for di, si, b in ytab:
        if first[di]:
            out[di] = b * buf[si]
            first[di] = False
        else:
            out[di] = out[di] + b * buf[si]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
