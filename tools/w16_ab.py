"""Dev tool: the 64-cout single-layer tile on v_mfma_f32_32x32x16_f16 (conv_dense.hip's wide kernel, SS4K_MODEL_NO_W16) against the same tile on
v_mfma_f32_16x16x32_f16 (conv_w16.hip, default), whole networks, interleaved rounds in one process.
usage: python tools/w16_ab.py [frames=4] [rounds=3] [srvgg|rrdbnet|bsvd ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kinds = sys.argv[3:] or ["srvgg"]
ctx = _capi.Context(0)
for kind in kinds:
    if kind == "srvgg":
        flat = W.flatten(W.srvgg_table(1, num_feat=64, num_conv=32, upscale=4), W.srvgg_keys(32))
        mk = lambda fl: _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=4, num_feat=64, num_block=32, flags=fl), flat)
        x = torch.rand(n, 3, 720, 1280, device="cuda")
    elif kind == "rrdbnet":
        flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
        mk = lambda fl: _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, flags=fl), flat)
        x = torch.rand(n, 3, 720, 1280, device="cuda")
    else:
        flat = W.flatten(W.bsvd_table(3), W.bsvd_keys())
        mk = lambda fl: _capi.Model(ctx, _capi.make_desc(_capi.BSVD, _capi.F16, scale=1, flags=fl), flat)
        x = torch.rand(n, 4, 720, 1280, device="cuda")
    ms = {"32x32x16 (wide)": mk(_capi.MODEL_NO_W16), "16x16x32 (w16)": mk(0)}
    for m in ms.values():
        for _ in range(6): m(x)
    torch.cuda.synchronize()
    iters = 10 if kind != "bsvd" else 30
    for r in range(rounds):
        for k, m in ms.items():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(iters): m(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
            print(f"{kind} round {r} {k:16s}: {1000 * dt:.3f} ms per {n} frames = {n / dt:.1f} frames/s", flush=True)
    del ms
