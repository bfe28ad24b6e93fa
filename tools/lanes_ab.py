"""Dev tool: frame lanes A/B (SS4K_LANES = concurrent launch chains per job) on the headline workload.
Interleaved rounds in ONE process on one device; the lanes builds must give bit-identical frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi, weights as W

variants = (sys.argv[1] if len(sys.argv) > 1 else "1,2,0").split(",")
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = _capi.Context(0)
flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
ups = {}
for v in variants:
    # variant syntax: <mode>[g<share>]: mode 0 = measured choice, 1 = one chain, 2 = two chains; g = grid share of a lane's launch
    m = v.split("m")
    os.environ["SS4K_RS_MASK"] = m[1] if len(m) > 1 else "32"   # layer shapes on the register-stationary kernel (models.cpp rs_shape_bit)
    g = m[0].split("g")
    os.environ["SS4K_LANES"] = g[0]
    os.environ["SS4K_LANE_GRID"] = g[1] if len(g) > 1 else "1.0"
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    ups[v] = (_capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0), sr)
frames = torch.from_numpy(np.random.default_rng(1000).integers(0, 256, (batch, 720, 1280, 3), dtype=np.uint8)).cuda()
outs = {v: torch.empty((batch, 1440, 2560, 3), dtype=torch.uint8, device="cuda") for v in variants}
for v in variants:
    for _ in range(6):
        ups[v][0](frames, outs[v])
torch.cuda.synchronize()
ref = outs[variants[0]]
for v in variants[1:]:
    print(f"lanes={v}: identical to lanes={variants[0]}: {bool(torch.equal(ref, outs[v]))}", flush=True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ups[v][0](frames, outs[v])
        torch.cuda.synchronize()
        res[v].append(10 * batch / (time.perf_counter() - t0))
for v in variants:
    print(f"lanes={v} batch={batch}: fps median {np.median(res[v]):.2f}  max {max(res[v]):.2f}  all {[round(x, 1) for x in res[v]]}", flush=True)
