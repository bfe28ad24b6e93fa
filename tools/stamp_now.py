import os, sys
sys.path.insert(0, os.getcwd())
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for name, c0, c1, co in [("conv1 64->32", 64, 0, 32), ("conv2 96->32", 64, 32, 32), ("conv4 160->32", 64, 96, 32), ("trunk 64->64", 64, 0, 64)]:
    for n in (2, 4):
        gf = 2 * 9 * (c0 + c1) * co * H * W * n / 1e9
        us0 = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, 0, 20)
        print(f"{name} n={n}: {us0:.1f} us {gf/us0*1e3:.0f} TFLOP/s", flush=True)
        us = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, 32, 20)
        print(f"   (stamped build: {us:.1f} us)", flush=True)
