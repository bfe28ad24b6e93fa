"""Dev probe (GPU box): does a service instance created after others run slower?  (bench.py: the `pipeline` leg lost 6-8 % whenever one more
service had been created before it.)  One process: pipeline service A, measure; an RRDBNet service, run and drop it; pipeline service B, measure;
A again."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from tests.helpers import smooth_u8
fr = torch.from_numpy(smooth_u8(1, (4, 720, 1280, 3))).cuda()
def pipeline():
    s = HipUpscalerService(device=0, upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=True, single_mode=True, weights="synthetic", seed=0, dtype="f16")
    s.proc_init(); return s
def rrdb(flags=0):
    s = HipUpscalerService(device=0, upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=0, dtype="f16", model_flags=flags)
    s.proc_init(); return s
def fps(s, reps=12):
    for _ in range(8): s.upscale(fr, wait=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): s.upscale(fr, wait=False)
    torch.cuda.synchronize(); return 4 * reps / (time.perf_counter() - t0)
main = rrdb(); print(f"main rrdbnet service:            {fps(main):6.1f}", flush=True)
A = pipeline(); print(f"pipeline A (2nd service):        {fps(A):6.1f}", flush=True)
x = rrdb(2); print(f"extra rrdbnet service (one chain): {fps(x):6.1f}", flush=True)
del x; torch.cuda.empty_cache()
B = pipeline(); print(f"pipeline B (after the extra one): {fps(B):6.1f}", flush=True)
print(f"pipeline A again:                {fps(A):6.1f}", flush=True)
print(f"pipeline B again:                {fps(B):6.1f}", flush=True)
C = pipeline(); print(f"pipeline C:                      {fps(C):6.1f}", flush=True)
print(f"main again:                      {fps(main):6.1f}", flush=True)
