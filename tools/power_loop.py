"""Dev tool: run one SRVGG x4 720p model variant in a loop for some seconds (model flags and seconds from argv) - for sampling rocm-smi
beside it (tools/power_sample.sh).  usage: python tools/power_loop.py <flags> <seconds>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W
fl = int(sys.argv[1]); secs = float(sys.argv[2])
ctx = _capi.Context(0)
flat = W.flatten(W.srvgg_table(1, num_feat=64, num_conv=32, upscale=4), W.srvgg_keys(32))
m = _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=4, num_feat=64, num_block=32, flags=fl), flat)
x = torch.rand(4, 3, 720, 1280, device="cuda")
for _ in range(5): m(x)
torch.cuda.synchronize()
print("START", flush=True)
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(10): m(x)
    torch.cuda.synchronize(); n += 10
dt = time.perf_counter() - t0
print(f"flags {fl}: {4 * n / dt:.1f} frames/s", flush=True)
