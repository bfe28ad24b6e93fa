import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for rep in range(2):
    for fl, what in ((0, "lrelu, no residual"), (2048, "x*0.2 + res1 (conv5 form)")):
        us = ctx.bench_conv(_capi.F16, 64, 128, 64, 4, H, W, fl, 30)
        print(f"conv5 n=4 {what}: {us:.1f} us", flush=True)
        ctx.bench_conv(_capi.F16, 64, 128, 64, 4, H, W, fl | 32, 3)
