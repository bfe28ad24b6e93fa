"""Dev tool: determinism soak of the round-3 / round-4 kernels - RRDBNet with the fused dense-block pairs and the single-layer wide
kernel (round 4), BSVD with the fused layer pairs, FSRCNN in fp16 mode, SRVGG with the fp16
HR tensor, the RRDBNet chain kernel - each job repeated N times on ragged and full sizes; every repeat must reproduce the
first output bit for bit (a hand-off or LDS-DMA race shows up as a differing frame).  usage: python tools/soak.py [repeats=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = _capi.Context(0)
g = torch.Generator().manual_seed(1)
jobs = []
bs = factory.build_denoise_model(ctx, weights="synthetic", dtype="f16")
for shape in ((4, 4, 720, 1280), (3, 4, 100, 252), (1, 4, 64, 60)):
    jobs.append((f"bsvd fused pairs {shape}", bs, torch.rand(*shape, generator=g).cuda()))
fs = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2), dtype="f16")
for shape in ((12, 1, 720, 1280), (3, 1, 150, 333)):
    jobs.append((f"fsrcnn f16 {shape}", fs, torch.rand(*shape, generator=g).cuda()))
fs32 = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2))
for shape in ((12, 1, 720, 1280), (3, 1, 97, 130)):
    jobs.append((f"fsrcnn fp32-grade (split MFMA head) {shape}", fs32, torch.rand(*shape, generator=g).cuda()))
# the default route (fused pairs with the ring table, 16x16x32 tile for conv5 / trunk / conv_hr), two launch chains and one, ragged and full size
rd = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=6), W.flatten(W.rrdbnet_table(6, scale=2, num_block=6), W.rrdbnet_keys(6)))
for shape in ((4, 3, 720, 1280), (1, 3, 720, 1280), (3, 3, 250, 330), (2, 3, 70, 66)):
    jobs.append((f"rrdbnet fused pairs {shape}", rd, torch.rand(*shape, generator=g).cuda()))
sv = factory.build_model_esrgan(ctx, "realesr-general-x4v3", weights="synthetic", dtype="f16", seed=3)
up = _capi.Upscaler(ctx, sv, (180, 320), (360, 640), True, False, None, 0.5)
fr = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (4, 180, 320, 3), dtype=np.uint8)).cuda()
jobs.append(("srvgg service, fp16 HR tensor (4, 180, 320, 3)", up, fr))
bad = 0
for name, fn, x in jobs:
    ref = fn(x).clone(); torch.cuda.synchronize()
    t0 = time.perf_counter(); diff = 0
    for i in range(reps):
        y = fn(x)
        if not torch.equal(y, ref): diff += 1
    torch.cuda.synchronize()
    print(f"{name}: {reps} repeats, {diff} differing, {1000 * (time.perf_counter() - t0) / reps:.2f} ms each", flush=True)
    bad += diff
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
