"""Dev tool: BSVD-32 on n frames of 720p with the fused inc / outc pairs (conv_pair.hip) and with two launches per pair
(SS4K_MODEL_NO_PAIR), interleaved rounds in one process.  usage: python tools/bsvd_ab.py [frames=4] [rounds=3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi
from sharkshark4k_amd.upscale import model as factory
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _capi.Context(0)
ms = {"fused pairs": factory.build_denoise_model(ctx, weights="synthetic", dtype="f16"),
      "two launches": factory.build_denoise_model(ctx, weights="synthetic", dtype="f16", flags=_capi.MODEL_NO_PAIR)}
x = torch.rand(n, 4, 720, 1280, device="cuda")
for m in ms.values():
    for _ in range(6): m(x)
torch.cuda.synchronize()
for r in range(rounds):
    for k, m in ms.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"round {r} {k:13s}: {1000 * dt:.3f} ms per {n} frames = {1000 * dt / n:.3f} ms/frame", flush=True)
