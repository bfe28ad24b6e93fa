"""Dev tool: 1-frame (and n-frame) RRDBNet x2 720p jobs under model-description flags, interleaved rounds in one process.
usage: python tools/n1_ab.py [frames=1] [rounds=3] [flags[:VAR=VAL[+VAR=VAL]],...]   (flags: integers, include/ss4k.h
SS4K_MODEL_*; VAR=VAL: dev-library switches read per forward, e.g. 128:SS4K_CHAIN_ABL=1 with SS4K_LIB=.../libss4k_hip_dev.so)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", str(0)]
def setenv(v):
    for k in [k for k in os.environ if k.startswith("SS4K_CHAIN_")]:
        del os.environ[k]
    for kv in v.split(":")[1].split("+") if ":" in v else []:
        k, val = kv.split("="); os.environ[k] = val
ctx = _capi.Context(0)
flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
ups = {}
for fl in variants:
    setenv(fl)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, flags=int(fl.split(":")[0])), flat)
    ups[fl] = (_capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0), sr)
frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (nf, 720, 1280, 3), dtype=np.uint8)).cuda()
out = torch.empty((nf, 1440, 2560, 3), dtype=torch.uint8, device="cuda")
ref = None
for fl in variants:
    setenv(fl)
    for _ in range(6):
        ups[fl][0](frames, out)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    else:
        print(f"flags {fl}: output equal to flags {variants[0]}: {torch.equal(ref, out)}", flush=True)
for r in range(rounds):
    for fl in variants:
        setenv(fl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ups[fl][0](frames, out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"round {r} flags {fl:>28s}: {20 * nf / dt:7.2f} fps  ({1000 * dt / 20:.3f} ms per job)", flush=True)
