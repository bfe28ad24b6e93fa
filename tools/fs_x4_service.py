"""Dev tool (GPU box): the reference's OWN FSRCNN configuration - the x4 network (the service always builds FSRCNN(4), fsrcnn_upscaler.py:101,
fsrcnn/factory.py:6) on 720p frames, then its always-bicubic resize to output_shape = 1440p (pipeline.py:46-50) - through
HipUpscalerService.upscale, 4-frame jobs, both arithmetic modes.  SS4K_LIB selects the library build (A/B: run once per build)."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (4, 720, 1280, 3), dtype=np.uint8)).cuda()
for dt in ("f16", "f32"):
    for out_shape in ((1440, 2560), None):
        svc = HipUpscalerService(device=0, upscaler_model="fsrcnn", scale=4, denoising=False, weights="synthetic", seed=0, fsrcnn_dtype=dt)
        svc.output_shape = out_shape
        svc.proc_init()
        for _ in range(60): svc.upscale(frames, wait=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(150): o = svc.upscale(frames, wait=False)
        torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        print(f"FSRCNN x4 720p {dt} -> {tuple(o.shape[1:3])}: {600 / dt_s:.1f} frames/s  crc {int(o.to(torch.int64).sum())}", flush=True)
        del svc
