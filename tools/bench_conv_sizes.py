import os, sys
sys.path.insert(0, os.getcwd())
import sharkshark4k_amd
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
for (cin, cout, n, h, w) in [(64, 32, 2, 1440, 2560), (64, 64, 2, 1440, 2560), (64, 32, 2, 360, 640), (64, 32, 32, 360, 640)]:
    for fl in (0, 4096):
        try:
            us = ctx.bench_conv(_capi.F16, cin, 0, cout, n, h, w, flags=fl, iters=10)
            gf = 2 * 9 * cin * cout * n * h * w / 1e9
            print(f"{cin}->{cout} n={n} {h}x{w} flags={fl}: {us:8.1f} us  {gf / us / 1e3 * 1e3:7.1f} TFLOP/s  {(cin + cout) * 2 * n * h * w / us / 1e6:6.2f} TB/s (in+out bytes)", flush=True)
        except Exception as e:
            print(cin, cout, fl, "error", str(e)[:100])
