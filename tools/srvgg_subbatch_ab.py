"""Dev tool (GPU box, dev library): does the SRVGG job (32 x conv 64->64 at 720p: a layer reads and writes 472 MB for four frames, more than the 256 MB
Infinity Cache) gain from going through the network fewer frames at a time?  SS4K_SUBBATCH = frames per pass, SS4K_LANES = launch chains.
usage: SS4K_LIB=.../libss4k_hip_dev.so python3 tools/srvgg_subbatch_ab.py [workload] [rounds]"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
from sharkshark4k_amd import _capi
wl = sys.argv[1] if len(sys.argv) > 1 else "srvgg"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = ["SS4K_LANES=0,SS4K_SUBBATCH=0", "SS4K_LANES=1,SS4K_SUBBATCH=0", "SS4K_LANES=2,SS4K_SUBBATCH=0", "SS4K_LANES=1,SS4K_SUBBATCH=1",
            "SS4K_LANES=1,SS4K_SUBBATCH=2", "SS4K_LANES=2,SS4K_SUBBATCH=2"]
frames = B.synthetic_frames(4, (720, 1280), 1000).to("cuda")
svcs, outs = {}, {}
for v in variants:
    for kv in v.split(","):
        k, val = kv.split("="); os.environ[k] = val
    svcs[v], _ = B.build_service(wl, 0, flags=_capi.MODEL_HR_F32)   # (a split batch needs the fp32 HR tensor: every variant runs that way)
    for _ in range(8): o = svcs[v].upscale(frames)
    torch.cuda.synchronize(); outs[v] = o.clone()
for v in variants[1:]:
    print(f"{v}: identical to {variants[0]}: {bool(torch.equal(outs[variants[0]], outs[v]))}", flush=True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): svcs[v].upscale(frames, wait=False)
        torch.cuda.synchronize(); res[v].append(80 / (time.perf_counter() - t0))
for v in variants:
    print(f"{wl} {v}: fps median {np.median(res[v]):.1f}  all {[round(x, 1) for x in res[v]]}", flush=True)
