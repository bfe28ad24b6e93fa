"""Dev tool: timing-only ablations of the conv kernel's memory traffic (see conv_mfma.hip launch_dbg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for rep in range(2):
  for name, c0, c1, co in [("conv4", 64, 96, 32), ("conv5", 64, 128, 64)]:
    gf = 2 * 9 * (c0 + c1) * co * H * W * 4 / 1e9
    for fl, what in ((32, "production + stamps"), (49, "no stores"), (48, "tiles from a 2 MB L2 window"),
                     (33, "L2 window + no stores"), (34, "every DMA from one hot line")):
        us = ctx.bench_conv(_capi.F16, c0, c1, co, 4, H, W, fl, 20)
        print(f"{name} n=4 {what}: {us:.1f} us {gf/us*1e3:.0f} TFLOP/s", flush=True)
