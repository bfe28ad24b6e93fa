"""Dev tool: RDB conv5 (192 -> 64, x 0.2 + x) in isolation on random operands, 4 frames of 360 x 640: the register-stationary kernel,
the wide LDS-weights kernel with its residual read from memory, and the same layer WITHOUT a residual (what the residual costs).
usage: SS4K_LIB=.../libss4k_hip_dev.so python tools/conv5_routes.py [rounds=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ctx = _capi.Context(0)
F16 = _capi.F16
cases = [("rs kernel, + x via the matrix core", 2048 | 4096), ("wide kernel, + x read from memory", 2048), ("wide kernel, no residual (LeakyReLU instead)", 0)]
gf = 2 * 9 * 192 * 64 * 4 * 360 * 640 / 1e9
for r in range(rounds):
    for name, fl in cases:
        us = ctx.bench_conv(F16, 64, 128, 64, 4, 360, 640, flags=fl, iters=40)
        print(f"round {r} {name:48s}: {us:7.1f} us = {gf / us * 1e-3 * 1e3:6.0f} TFLOP/s", flush=True)
