"""Dev tool: per-step times of the headline job with frame lanes on / off (is the gain stable from step to step?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi, weights as W
ctx = _capi.Context(0)
flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
frames = torch.from_numpy(np.random.default_rng(1000).integers(0, 256, (4, 720, 1280, 3), dtype=np.uint8)).cuda()
out = torch.empty((4, 1440, 2560, 3), dtype=torch.uint8, device="cuda")
for v in sys.argv[1].split(","):
    os.environ["SS4K_LANES"] = v
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    up = _capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0)
    for _ in range(3):
        up(frames, out)
    torch.cuda.synchronize()
    ts = []
    for i in range(40):
        t0 = time.perf_counter(); up(frames, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for i in range(30):
        up(frames, out)
    torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 30 * 1e3
    print(f"lanes={v}: per-step ms min {min(ts):.2f} median {np.median(ts):.2f} max {max(ts):.2f}; 30 back-to-back steps {tb:.2f} ms/step = {4000 / tb:.1f} fps")
    print("   ", " ".join(f"{t:.1f}" for t in ts), flush=True)
    del up, sr
