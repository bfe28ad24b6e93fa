"""Dev probe (GPU box): small-job overlap over the service's job sets - how many sets, which job sizes; host enqueue time and frames/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from tests.helpers import smooth_u8
KW = dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=0, dtype="f16")
fr = torch.from_numpy(smooth_u8(1, (4, 720, 1280, 3))).cuda()
junk = [torch.cuda.Stream() for _ in range(5)]   # a process that has used other streams before (bench.py, a test suite): hardware queues are shared
for st in junk:
    with torch.cuda.stream(st): torch.zeros(1, device="cuda")
def run(label, fn, frames_per_call, reps=40):
    for _ in range(6): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label:72s} host enqueue {1e3 * (t1 - t0) / reps:6.2f} ms/job, {frames_per_call * reps / (t2 - t0):6.1f} frames/s", flush=True)
for sets, maxf, n in ((2, 1, 1), (3, 1, 1), (4, 1, 1), (2, 2, 2), (2, 4, 4), (2, 1, 4), (2, 1, 2)):
    svc = HipUpscalerService(device=0, overlap_sets=sets, overlap_max_frames=maxf, **KW); svc.proc_init()
    x = fr[:n]
    run(f"{n}-frame jobs, {sets} job sets, jobs of <= {maxf} frames alternate", lambda: svc.upscale(x, wait=False), n)
    del svc; torch.cuda.empty_cache()
svc = HipUpscalerService(device=0, overlap_jobs=False, **KW); svc.proc_init()
for n in (1, 2, 4):
    x = fr[:n]
    run(f"{n}-frame jobs, overlap_jobs=False (one set, current stream)", lambda: svc.upscale(x), n)
