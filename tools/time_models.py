"""Dev tool: time whole-network forwards (ms per call) on the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory

ctx = _capi.Context(0)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
bs = factory.build_denoise_model(ctx, weights="synthetic", dtype="f16")
for n in (1, 4):
    x = torch.rand(n, 4, 720, 1280, device="cuda")
    ms = timeit(lambda: bs(x))
    print(f"BSVD-32 720p n={n}: {ms:.2f} ms/call, {0.544*n/ms:.0f} TFLOP/s", flush=True)
sv = factory.build_model_esrgan(ctx, "realesr-general-x4v3", weights="synthetic", dtype="f16")
x = torch.rand(4, 3, 720, 1280, device="cuda")
ms = timeit(lambda: sv(x), 5)
print(f"SRVGG x4 720p n=4: {ms:.2f} ms/call, {2.228*4/ms:.0f} TFLOP/s", flush=True)
