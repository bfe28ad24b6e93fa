"""Per-layer-shape A/B of the two fp16 conv kernels on one device, interleaved rounds in one process
(cdna_hip_programming.md rule 24): LDS-weights kernel (conv_mfma.hip, 32x32x16 MFMA, two workgroups per CU)
vs register-stationary kernel (conv_rs.hip, 16x16x32 MFMA, one workgroup per CU).
usage: SS4K_LIB=<pkg>/libss4k_hip_dev.so python tools/conv_sweep.py [frames] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SS4K_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sharkshark-4k_amd", "libss4k_hip_dev.so"))
import numpy as np
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _capi.Context(0)
H, W = 360, 640
shapes = [("conv1  64->32", 64, 0, 32, 0), ("conv2  96->32", 64, 32, 32, 0), ("conv3 128->32", 64, 64, 32, 0),
          ("conv4 160->32", 64, 96, 32, 0), ("conv5 192->64 (+x)", 64, 128, 64, 2048), ("trunk  64->64", 64, 0, 64, 0)]
print(f"{n} frames of {H}x{W}, median of {rounds} interleaved rounds x 20 launches")
for name, c0, c1, co, fl in shapes:
    t = {0: [], 4096: [], 4096 + 8192: []}
    for r in range(rounds):
        for rs in (0, 4096, 4096 + 8192):
            t[rs].append(ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, flags=fl | rs, iters=20))
    gf = 2 * 9 * (c0 + c1) * co * n * H * W / 1e9
    a, b, c = float(np.median(t[0])), float(np.median(t[4096])), float(np.median(t[4096 + 8192]))
    print(f"{name:20s} LDS-weights {a:7.1f} us = {gf / a * 1e3:6.0f} TFLOP/s | register-stationary {b:7.1f} us = {gf / b * 1e3:6.0f} TFLOP/s | x{a / b:.2f}"
          f" | eight-wave RS {c:7.1f} us = {gf / c * 1e3:6.0f} TFLOP/s | x{a / c:.2f}")
