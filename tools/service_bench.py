"""Throughput through the drop-in boundary itself: HipUpscalerService behind the reference's queue
API (worker process, UpscalerQueueEntry with device tensors), 4-frame 720p jobs, RRDBNet x2 fp16.
usage: python tools/service_bench.py [jobs] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if __name__ == "__main__":
    import sharkshark4k_amd  # noqa: F401
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
    from sharkshark4k_amd.util import Profiler
    jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    svc = HipUpscalerService(device=0, denoising=False, upscaler_model="realesrgan", model_name="RealESRGAN_x2plus",
                             scale=2, lr_level=3, dtype="f16", weights="synthetic")
    svc.output_shape = (1440, 2560)
    svc.start()
    try:
        frames = [torch.randint(0, 256, (4, 720, 1280, 3), dtype=torch.uint8, device="cuda") for _ in range(depth)]
        for f in frames: f.share_memory_()
        # warm up (first job builds the model in the worker)
        svc.push_job(UpscalerQueueEntry(frames=frames[0], step=-1, profiler=Profiler()), timeout=600)
        svc.get_result(timeout=600)
        t0 = time.time(); sent = got = 0
        while got < jobs:
            while sent < jobs and sent - got < depth:
                svc.push_job(UpscalerQueueEntry(frames=frames[sent % depth], step=sent, profiler=Profiler()), timeout=60)
                sent += 1
            r = svc.get_result(timeout=120)
            assert r.step == got and r.frames.shape == (4, 1440, 2560, 3)
            _ = r.frames[0, 0, 0, 0].item()  # consumer touches the result (sync)
            got += 1
        dt = time.time() - t0
        print(f"service boundary: {jobs} jobs x 4 frames in {dt:.2f} s = {4 * jobs / dt:.1f} frames/s "
              f"(worker-side upscale {1000 * r.elapsed:.1f} ms/job)")
    finally:
        svc.stop()
