"""Dev probe (GPU box): do 4-frame jobs gain from alternating over the job sets (overlap_max_frames=4) now that every stream is vetted?
usage: python3 tools/multiframe_sets_probe.py [sets]"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw, _, _ = B.SERVICE_OF["rrdbnet"]
frames = B.synthetic_frames(4, (720, 1280), 1000).to("cuda")
def make(**extra):
    s = HipUpscalerService(device=0, weights="synthetic", seed=0, dtype="f16", lr_shape=(720, 1280), **kw, **extra); s.proc_init(); return s
svcs = {"set 0 only": make(), f"alternating over {sets} sets": make(overlap_max_frames=4, overlap_sets=sets)}
for s in svcs.values():
    for _ in range(10): s.upscale(frames, wait=False)
    torch.cuda.synchronize()
res = {k: [] for k in svcs}
for r in range(4):
    for k, s in svcs.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(24): s.upscale(frames, wait=False)
        torch.cuda.synchronize(); res[k].append(96 / (time.perf_counter() - t0))
for k in svcs: print(f"4-frame jobs, {k}: median {np.median(res[k]):.1f}  all {[round(x, 1) for x in res[k]]}", flush=True)
