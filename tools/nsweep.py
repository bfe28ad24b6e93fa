import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for name, c0, c1, co in [("conv1", 64, 0, 32), ("conv2", 64, 32, 32), ("conv3", 64, 64, 32), ("conv4", 64, 96, 32), ("conv5", 64, 128, 64)]:
    row = []
    for n in (1, 2, 3, 4, 6, 8):
        us = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, 2048 if co == 64 else 0, 20)
        row.append(f"n={n}: {us/n:6.1f}")
    print(name, "us/frame  ", "  ".join(row), flush=True)
