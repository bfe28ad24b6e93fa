"""Ablation timing of single conv layers (dev tool; needs a GPU).
Variants are separate compile-time builds of the fp16 kernel: 0 production, 16 no epilogue,
28 MFMA+LDS only (no DMA, no epilogue), 18 DMA only (no MFMA, no epilogue), 32 phase stamps."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd import _capi  # noqa: E402

ctx = _capi.Context(0)
H, W = 360, 640
layers = [("conv1 64->32", 64, 0, 32), ("conv3 64+64->32", 64, 64, 32), ("conv5 64+128->64", 64, 128, 64),
          ("64->64", 64, 0, 64)]
flagsets = [("full(8-row tiles for 32-cout)", 0), ("full 16-row tiles", 64)]
for name, c0, c1, co in layers:
    gf = 2 * 9 * (c0 + c1) * co * H * W / 1e9
    row = []
    for fn, fl in flagsets:
        us = ctx.bench_conv(_capi.F16, c0, c1, co, 1, H, W, fl, 30)
        row.append(f"{fn}={us:.1f}")
    print(f"{name} [{gf:.1f} GFLOP] us: " + "  ".join(row) + f"  -> {gf/float(row[0].split('=')[1])*1e3:.0f} TFLOP/s", flush=True)
    ctx.bench_conv(_capi.F16, c0, c1, co, 1, H, W, 32, 3)
    ctx.bench_conv(_capi.F16, c0, c1, co, 1, H, W, 32 | 64, 3)
for n in (1, 4):
    us = ctx.bench_conv(_capi.F16, 64, 64, 32, n, H, W, 0, 30)
    print(f"conv3 batch {n}: {us:.1f} us -> {2*9*128*32*H*W*n/us/1e6:.0f} TFLOP/s")
for (h, w) in ((720, 1280), (1440, 2560)):
    us = ctx.bench_conv(_capi.F16, 64, 0, 64, 1, h, w, 0, 10)
    print(f"64->64 @{h}x{w}: {us:.1f} us -> {2*9*64*64*h*w/us/1e6:.0f} TFLOP/s")
