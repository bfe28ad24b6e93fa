#!/usr/bin/env python3
"""Where does a one-frame request's time go?  (GPU box; companion of tools/latency_probe.py.)  One request at a time through a spawned worker
whose on_queue copies the result to the host; wall-clock stamps at every hop, medians printed."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
from sharkshark4k_amd.util import Profiler


class Stamped(HipUpscalerService):
    def proc_job_recieved(self, job):
        job.profiler.set("t_recv", time.time())
        r = super().proc_job_recieved(job)
        r.profiler.set("t_enq", time.time())
        return r

    def _deliver(self, entry):
        entry.profiler.set("t_deliver", time.time())
        super()._deliver(entry)


class HostCopyOnQueue:
    def __init__(self, svc, mode):
        self.q, self.mode = svc.result_queue, mode

    def __call__(self, entry):
        p = entry.profiler
        p.set("t_onq", time.time())
        if self.mode == "cpu":
            f = entry.frames.cpu(); p.set("t_cpu", time.time())
            f = f.clone(); p.set("t_clone", time.time())
        elif self.mode == "sync":
            torch.cuda.current_stream().synchronize(); p.set("t_cpu", time.time()); p.set("t_clone", time.time())
            f = entry.frames
        elif self.mode.startswith("sync_"):
            torch.cuda.current_stream().synchronize(); p.set("t_cpu", time.time())
            n = 1440 * 2560 * 3
            if self.mode == "sync_malloc":       # a fresh 11 MB host allocation, touched and freed; the result itself stays on the device
                t = torch.empty(n, dtype=torch.uint8); t.fill_(1); del t
                f = entry.frames
            elif self.mode == "sync_small":      # a small host tensor through the queue
                f = torch.zeros(16, dtype=torch.uint8)
            elif self.mode == "sync_const":      # the SAME shared 11 MB host tensor every time (no new mapping in the worker)
                if not hasattr(self, "const"):
                    self.const = torch.zeros(n, dtype=torch.uint8).share_memory_()
                f = self.const
            else:                                # sync_new11: a new 11 MB host tensor through the queue every time (no device copy involved)
                f = torch.zeros(n, dtype=torch.uint8)
            p.set("t_clone", time.time())
        else:
            f = entry.frames; p.set("t_cpu", time.time()); p.set("t_clone", time.time())
        self.q.put(UpscalerQueueEntry(frames=f, step=entry.step, profiler=p))
        p.set("t_put", time.time())


if __name__ == "__main__":
    parent_cpu = "--parent-cpu" in sys.argv      # the requesting process never touches the GPU (frames travel as host tensors)
    modes = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cpu", "sync", "ipc"]
    for mode in modes:
        for overlap in (False,):
            svc = Stamped(lr_level=3, device=0, denoising=False, upscaler_model="realesrgan", batch_size=1, jit_mode=False, lr_hr_resize=False,
                          model_name="RealESRGAN_x2plus", weights="synthetic", seed=0, dtype="f16", overlap_jobs=overlap)
            svc.on_queue = HostCopyOnQueue(svc, mode)
            svc.start()
            frame = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8))
            if not parent_cpu:
                frame = frame.cuda()
            rows = []
            for i in range(60):
                p = Profiler(); t0 = time.time()
                svc.push_job(UpscalerQueueEntry(frames=frame, step=i, profiler=p), timeout=600)
                r = svc.get_result(timeout=600)
                if r.frames.is_cuda:
                    r.frames.cpu()
                t1 = time.time()
                d = r.profiler.data
                if i >= 30:
                    rows.append([d["t_recv"] - t0, d["t_enq"] - d["t_recv"], d["t_deliver"] - d["t_enq"], d["t_onq"] - d["t_deliver"], d["t_cpu"] - d["t_onq"],
                                 d["t_clone"] - d["t_cpu"], t1 - d["t_clone"], t1 - t0])
            svc.stop()
            m = np.median(np.array(rows), 0) * 1000
            print(f"mode={mode:10} parent_on_gpu={not parent_cpu!s:5} overlap={overlap!s:5}: push->recv {m[0]:6.2f}  enqueue {m[1]:6.2f}  held {m[2]:6.2f}  before_deliver {m[3]:6.2f}  .cpu()/sync {m[4]:6.2f}  clone {m[5]:6.2f}  "
                  f"put->parent has it {m[6]:6.2f}  total {m[7]:6.2f} ms", flush=True)
