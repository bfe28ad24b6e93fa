import os, sys
sys.path.insert(0, os.getcwd())
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
shapes1 = {1: "<1,2,4> 8 rows, 3 WG/CU", 2: "<1,4,4> 16 rows, 2 WG/CU", 3: "<1,2,8> 16 rows, 1 WG/CU", 4: "<1,4,8> 32 rows, 1 WG/CU"}
shapes2 = {1: "<2,2,8> 16 rows, 1 WG/CU", 2: "<2,4,4> 16 rows, 2 WG/CU", 3: "<2,4,8> 32 rows, 1 WG/CU", 4: "<2,2,4> 8 rows, 2 WG/CU"}
stamps = len(sys.argv) > 1
for rep in range(2):
  for name, c0, c1, co in [("conv1", 64, 0, 32), ("conv4", 64, 96, 32), ("conv5", 64, 128, 64)]:
    for n in (4,):
        gf = 2 * 9 * (c0 + c1) * co * H * W * n / 1e9
        for sid, what in (shapes1 if co == 32 else shapes2).items():
            us = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, (sid << 8) | (32 if stamps else 0), 20)
            print(f"{name} n={n} {what}: {us:.1f} us {gf/us*1e3:.0f} TFLOP/s", flush=True)
