#!/usr/bin/env python3
"""Prints one line per case: a SHA-256 of the FSRCNN network's output tensor (matrix-core modes), for shapes whose bands straddle one or
many plane boundaries.  tests/test_gpu_parity.py runs it twice in child processes - as built (tall bands over the stacked planes) and
with SS4K_MH_NO_TALL / SS4K_TAIL_NO_TALL (the classic whole-bands-per-plane grids; the library reads the switches once per process) -
and compares the lines: the two grids walk the same rows through the same arithmetic, so every byte must agree."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd import _capi  # noqa: E402
from sharkshark4k_amd.upscale import model as factory  # noqa: E402
from sharkshark4k_amd import weights as W  # noqa: E402

#: (factor, planes, h, w): one band over seven planes; bands of 33-35 rows over planes of 33 / 70; a plane shorter than a tail band;
#: configs[1]'s own 12 x 720 x 1280
CASES = [(2, 7, 5, 40), (4, 5, 9, 31), (2, 5, 33, 300), (2, 3, 70, 141), (4, 2, 45, 66), (2, 12, 64, 260), (2, 12, 720, 1280), (4, 3, 180, 320)]


def main():
    ctx = _capi.Context(0)
    for factor, planes, h, w in CASES:
        table = W.fsrcnn_table(seed=factor)
        x = torch.rand(planes, 1, h, w, generator=torch.Generator().manual_seed(h * 1000 + w)).cuda()
        for dtype in ("f16", "f32"):
            m = factory.build_model_fsrcnn(ctx, factor=factor, weights=table, dtype=dtype)
            y = m(x).float().cpu().contiguous()
            assert torch.isfinite(y).all()
            print(f"x{factor} {planes}x{h}x{w} {dtype} {hashlib.sha256(y.numpy().tobytes()).hexdigest()}")
    print("FS BANDS DONE")


if __name__ == "__main__":
    main()
