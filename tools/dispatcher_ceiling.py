#!/usr/bin/env python3
"""How many jobs per second can ONE Python parent push through the node's dispatcher?  (CPU only; no GPU involved.)

G no-op workers - real spawned ``HipUpscalerService`` worker processes with the real queues, ``HostFrames`` descriptors, pinned-ring slot
accounting and ordered fan-in; only the device work is replaced by nothing (the worker answers at once, the result slot keeps whatever
it held) - are fed 720p frames from host memory by ``UpscalerNode.submit_batch`` and drained by ``poll()``.  What is timed is therefore
exactly the host-side cost the parent pays per job: one 2.76 MB-per-frame copy into the input ring, two queue hops of ~ 100-byte
descriptors, the step re-ordering, a zero-copy view of the 11 MB-per-frame result.  The GPUs need 8 x 130 = 1040 frames/s at four-frame
jobs and 8 x 125 = 1000 at one-frame jobs; the dispatcher has to stay above that.

usage: python tools/dispatcher_ceiling.py [--workers 8] [--seconds 3] [--json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd.node import UpscalerNode  # noqa: E402
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService  # noqa: E402


class NoopWorker(HipUpscalerService):
    """The worker with its device stages removed (NOT a product path): no context, no model, a host job is answered at once."""

    def _open_device(self):
        self.torch_device = torch.device("cpu")
        self._local_device = None

    def _build_models(self):
        self.model = self.denoise_model = None

    def _init_job_sets(self):
        self.deliver_lag = 0
        self._pending = {}

    def _host_job(self, hf):
        oh, ow = self.out_hw(hf.shape[1], hf.shape[2])
        self.host_rings[1].view(hf.out_slot, (1,))[0] = hf.shape[0]     # (one byte: the result slot was written by this job)
        return (hf.shape[0], oh, ow, 3), None


def measure(workers: int, job_frames: int, seconds: float, slots: int = 6):
    fps = job_frames if job_frames < 4 else 24          # small_batch_size = min(4, fps)
    node = UpscalerNode(devices=list(range(workers)), service_cls=NoopWorker, backend="gloo", fps=fps, frame_skips=False, host_slots=slots,
                        upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", lr_shape=(720, 1280))
    node.start(timeout=300)
    try:
        batch = np.random.default_rng(0).integers(0, 256, (job_frames * workers, 720, 1280, 3), dtype=np.uint8)   # one job per worker and call
        done = frames = 0
        submitted = 0
        t_end = time.perf_counter() + 0.5
        t0 = None
        inflight_cap = workers * 3
        while True:
            now = time.perf_counter()
            if t0 is None and now >= t_end:      # warm-up over
                t0, done, frames, t_end = now, 0, 0, now + seconds
            elif t0 is not None and now >= t_end:
                break
            if submitted - node.dispatcher.next_emit < inflight_cap:
                submitted += len(node.submit_batch(batch))
            for e in node.poll(0.0):
                assert e.frames.shape == (job_frames, 1440, 2560, 3) and int(e.frames.reshape(-1)[0]) == job_frames
                done += 1
                frames += job_frames
        dt = time.perf_counter() - t0
        rep = node.report()
        assert rep["lost"] == 0 and rep["dropped"] == 0 and rep["host_fallback"] == 0, rep
        return {"workers": workers, "job_frames": job_frames, "jobs_per_s": done / dt, "frames_per_s": frames / dt,
                "host_copy_GBps": frames * 720 * 1280 * 3 / dt / 1e9}
    finally:
        node.stop()
        node.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = [measure(a.workers, jf, a.seconds) for jf in (4, 1)]
    if a.json:
        print(json.dumps(rows))
    else:
        print(f"dispatcher ceiling, {a.workers} no-op workers, 720p in / 1440p out, {os.cpu_count()} host threads:")
        for r in rows:
            need = a.workers * (130 if r["job_frames"] == 4 else 125)
            print(f"  {r['job_frames']}-frame jobs: {r['jobs_per_s']:8.1f} jobs/s = {r['frames_per_s']:8.1f} frames/s "
                  f"(needed for {a.workers} GPUs: {need}; parent copies {r['host_copy_GBps']:.2f} GB/s into the rings)")
