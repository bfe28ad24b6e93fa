"""Dev tool: 1-frame jobs (image-server mode) from ONE caller stream vs from TWO caller streams (two upscalers, one context
each, jobs alternating) - what the frame lanes do inside a multi-frame job, done by the caller for independent 1-frame jobs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi, weights as W

flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
def make():
    ctx = _capi.Context(0)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    return ctx, sr, _capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (batch, 720, 1280, 3), dtype=np.uint8)).cuda()
a, b = make(), make()
outs = [torch.empty((batch, 1440, 2560, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(6):
    a[2](frames, outs[0]); b[2](frames, outs[1])
torch.cuda.synchronize()
reps = 40
t0 = time.perf_counter()
for _ in range(reps):
    a[2](frames, outs[0])
torch.cuda.synchronize()
one = reps * batch / (time.perf_counter() - t0)
t0 = time.perf_counter()
for i in range(reps):
    with torch.cuda.stream(s1 if i % 2 == 0 else s2):
        (a if i % 2 == 0 else b)[2](frames, outs[i % 2])
torch.cuda.synchronize()
two = reps * batch / (time.perf_counter() - t0)
print(f"{batch}-frame jobs, one caller stream: {one:.1f} frames/s; two caller streams (two upscalers): {two:.1f} frames/s ({two / one:.3f}x)")
