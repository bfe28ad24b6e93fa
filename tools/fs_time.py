import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
m = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2))
x = torch.rand(12, 1, 720, 1280, device="cuda")
for _ in range(2): m(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): m(x)
e1.record(); torch.cuda.synchronize()
print("fsrcnn 12 planes 720p: %.2f ms" % (e0.elapsed_time(e1) / 5))
