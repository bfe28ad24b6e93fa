"""Dev tool: A/B of environment switches that the library reads when a model is built, on the headline job.
usage: python tools/env_ab.py "SS4K_MB5=0;SS4K_MB5=1" [batch] [rounds]   (variants separated by ';', several VAR=VAL per variant by ',')
Interleaved rounds in one process; the first variant is the reference of the bit-identity check."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi, weights as W
variants = sys.argv[1].split(";")
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = _capi.Context(0)
flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
def setenv(v):
    for kv in v.split(","):
        k, val = kv.split("=")
        os.environ[k] = val
frames = torch.from_numpy(np.random.default_rng(1000).integers(0, 256, (batch, 720, 1280, 3), dtype=np.uint8)).cuda()
ups, outs = {}, {}
for v in variants:
    setenv(v)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    ups[v] = (_capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0), sr)
    outs[v] = torch.empty((batch, 1440, 2560, 3), dtype=torch.uint8, device="cuda")
    for _ in range(6):
        ups[v][0](frames, outs[v])
    torch.cuda.synchronize()
for v in variants[1:]:
    print(f"{v}: identical to {variants[0]}: {bool(torch.equal(outs[variants[0]], outs[v]))}", flush=True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        setenv(v)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            ups[v][0](frames, outs[v])
        torch.cuda.synchronize()
        res[v].append(10 * batch / (time.perf_counter() - t0))
for v in variants:
    print(f"{v} batch={batch}: fps median {np.median(res[v]):.2f}  all {[round(x, 1) for x in res[v]]}", flush=True)
