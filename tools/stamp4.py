import os, sys
sys.path.insert(0, os.getcwd())
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for rep in range(2):
  for name, c0, c1, co in [("conv4", 64, 96, 32), ("conv5", 64, 128, 64)]:
    for n in (4,):
        gf = 2 * 9 * (c0 + c1) * co * H * W * n / 1e9
        for fl, what in ((32, "stamps"), (33, "tiles from a 2 MB L2 window, no stores"), (34, "DMA from one hot line"), (44, "no DMA")):
            us = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, fl, 20)
            print(f"{name} n={n} {what}: {us:.1f} us {gf/us*1e3:.0f} TFLOP/s", flush=True)
