"""Dev tool: what the epilogue of a 32-cout layer costs - the layer timed complete, without its stores (DBG_NO_STORE) and without
its epilogue (DBG_NO_EPILOGUE), per-launch sizes of the headline job (2 frames per lane launch) and 4 frames.
usage: SS4K_LIB=.../libss4k_hip_dev.so python tools/epi_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SS4K_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sharkshark-4k_amd", "libss4k_hip_dev.so"))
import numpy as np
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi
ctx = _capi.Context(0)
H, W = 360, 640
for name, c0, c1, co in [("conv1 64->32", 64, 0, 32), ("conv2 96->32", 64, 32, 32), ("conv4 160->32", 64, 96, 32)]:
    for n in (2, 4):
        t = {0: [], 1: [], 16: [], 17: []}
        for r in range(5):
            for fl in t:
                t[fl].append(ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, fl, 20))
        m = {k: float(np.median(v)) for k, v in t.items()}
        print(f"{name} n={n}: complete {m[0]:.1f} us | no stores {m[1]:.1f} ({100 * (m[0] - m[1]) / m[0]:.1f} %) | no epilogue arithmetic {m[16]:.1f} ({100 * (m[0] - m[16]) / m[0]:.1f} %) | neither {m[17]:.1f} ({100 * (m[0] - m[17]) / m[0]:.1f} %)", flush=True)
