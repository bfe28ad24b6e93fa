#!/usr/bin/env python3
"""Latency of one-frame requests through a spawned service worker, the image server's way (GPU box).

The image server (``src/sharkshark/image_server/image_pipeline.py:280-330``) has many request threads: each pushes a one-frame job and blocks
until ITS result arrives; the worker's ``on_queue`` copies every result to the host (``.cpu().clone()``, ``:38-47``) before it reads the
next job.  This tool does the same against ``HipUpscalerService`` (RRDBNet x2, 720p frame, the headline network) with c = 1, 2, 4, 8
requests in flight, for ``host_results`` on (the worker hands ``on_queue`` a pinned host copy made on its D2H stream) and off (``on_queue``
does a pageable ``.cpu()`` of a device tensor, the reference's code unchanged) and ``overlap_jobs`` on (one-frame jobs alternate over three job sets, results held until ready / until two more
jobs are enqueued) and off (one job set, every result leaves before the next job is read): p50 / p99 of submit -> result and the
throughput, per level.

usage: python tools/latency_probe.py [requests per level = 120]
"""
import os
import sys
import threading
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService  # noqa: E402
from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry  # noqa: E402
from sharkshark4k_amd.util import Profiler  # noqa: E402


class HostCopyOnQueue:
    """The image server's ``pipeline_onqueue``: inside the worker, copy the result to the host, then put it on the result queue."""

    def __init__(self, svc):
        self.q = svc.result_queue

    def __call__(self, entry):
        self.q.put(UpscalerQueueEntry(frames=entry.frames.cpu().clone(), audio_segment=None, step=entry.step, elapsed=entry.elapsed,
                                      last_modified=entry.last_modified, profiler=entry.profiler))


def run(overlap: bool, levels, n_req: int, host_results: bool = False):
    svc = HipUpscalerService(lr_level=3, device=0, denoising=False, denoise_rate=0.2, upscaler_model="realesrgan", batch_size=1, jit_mode=False,
                             lr_hr_resize=False, model_name="RealESRGAN_x2plus", weights="synthetic", seed=0, dtype="f16", overlap_jobs=overlap)
    if host_results:
        svc.host_results = True          # results arrive on the result queue as host tensors (views of the worker's pinned result ring)
    else:
        svc.on_queue = HostCopyOnQueue(svc)   # the reference's code: .cpu().clone() in the worker, a NEW host tensor through the queue
    svc.start()
    frame = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8)).cuda()
    waiting, lock = {}, threading.Lock()

    def reader():
        while True:
            e = svc.result_queue.get()
            if e is None:
                return
            with lock:
                ev = waiting.pop(e.step)
            int(e.frames[0, 0, 0, 0]); assert not e.frames.is_cuda   # (the result is on the host)
            ev[1] = time.perf_counter()
            ev[0].set()
    rd = threading.Thread(target=reader, daemon=True)
    rd.start()

    def request(step):
        ev = [threading.Event(), None]
        with lock:
            waiting[step] = ev
        t0 = time.perf_counter()
        svc.push_job(UpscalerQueueEntry(frames=frame, step=step, profiler=Profiler()), timeout=600)
        ev[0].wait(600)
        return ev[1] - t0

    for i in range(40):          # warm-up: builds the job sets, settles the launch-chain choice
        request(("w", i))
    rows = []
    for c in levels:
        lat = []

        def client(k):
            for i in range(n_req // c):
                lat.append(request((c, k, i)))
        t0 = time.perf_counter()
        th = [threading.Thread(target=client, args=(k,)) for k in range(c)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        a = np.sort(np.array(lat)) * 1000
        rows.append((c, len(a), float(np.percentile(a, 50)), float(np.percentile(a, 99)), float(a.max()), len(a) / dt))
    svc.result_queue.put(None)
    svc.stop()
    return rows


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    levels = (1, 2, 4, 8)
    print(f"one-frame 720p RRDBNet x2 requests through a spawned worker, on_queue copies each result to the host (image server): {n} requests per level")
    for host_results, overlap in ((True, True), (True, False), (False, True), (False, False)):
        for c, cnt, p50, p99, mx, fps in run(overlap, levels, n if host_results else min(n, 48), host_results):
            print(f"  host_results={host_results!s:5} overlap_jobs={overlap!s:5}  in flight {c}:  p50 {p50:7.2f} ms  p99 {p99:7.2f} ms  max {mx:7.2f} ms  {fps:6.1f} frames/s  ({cnt} requests)", flush=True)
