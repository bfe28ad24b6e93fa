"""Dev probe (GPU box): where does the in-process one-frame overlap go?  Variants of the alternating loop, host enqueue time and fps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from tests.helpers import smooth_u8
KW = dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=0, dtype="f16")
svc = HipUpscalerService(device=0, **KW); svc.proc_init()
fr = torch.from_numpy(smooth_u8(1, (1, 720, 1280, 3))).cuda()
def run(label, fn, reps=40):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{label:70s} host enqueue {1e3 * (t1 - t0) / reps:6.2f} ms/job, {reps / (t2 - t0):6.1f} fps", flush=True)
run("service.upscale(wait=False), alternating sets", lambda: svc.upscale(fr, wait=False))
keep = []
run("... results kept alive", lambda: keep.append(svc.upscale(fr, wait=False))); keep.clear()
svc.overlap_jobs = False
run("overlap_jobs=False (one set, current stream)", lambda: svc.upscale(fr))
svc.overlap_jobs = True
# raw: the two upscalers on two torch streams, preallocated outputs (round 4's two-callers arrangement)
ups = [svc._get_upscaler(0), svc._get_upscaler(1)]
sts = [torch.cuda.Stream(), torch.cuda.Stream()]
oh, ow = ups[0].out_shape(1, 720, 1280)
outs = [torch.empty((1, oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
i = [0]
def raw():
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(sts[k]): ups[k](fr, outs[k])
run("raw: two upscalers, two streams, preallocated outputs", raw)
def raw_alloc():
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(sts[k]): ups[k](fr)
run("raw + output allocated per job inside the stream context", raw_alloc)
def raw_wait():
    k = i[0] & 1; i[0] += 1
    sts[k].wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sts[k]): ups[k](fr, outs[k])
run("raw + side.wait_stream(current)", raw_wait)
def raw_ev():
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(sts[k]): ups[k](fr, outs[k])
    sts[k].record_event()
run("raw + side.record_event()", raw_ev)
def raw_rs():
    k = i[0] & 1; i[0] += 1
    with torch.cuda.stream(sts[k]): o = ups[k](fr)
    o.record_stream(torch.cuda.current_stream())
run("raw + alloc + out.record_stream(current)", raw_rs)
