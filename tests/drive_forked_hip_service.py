#!/usr/bin/env python3
"""A fresh interpreter that does what the reference's callers do, on a real GPU: build the service BEFORE anything touches the GPU,
``start()`` it - the worker is FORKED (``BaseService.start_method``: an untouched parent forks, as the reference does, so a callback
that is a bound method of an unpicklable-by-spawn pipeline object and a module global both reach the worker) -, and only then create
device tensors in the parent (``torch.tensor(img, device=upscaler.device)``, the image server's line) and push jobs made of the
CALLER's own record / profiler types.  Results come back through the second service of the pipeline object and are compared, byte for
byte, with an in-process upscaler built afterwards from the same generated weights.

Run by tests/test_gpu_callers.py as a child process (never imported by a process that already holds a HIP context).
Prints FORKED SERVICE OK.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd.upscale.base_service import gpu_runtime_touched  # noqa: E402
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService  # noqa: E402
from tests.caller_shapes import CallerEntry, CallerPipeline, CallerProfiler  # noqa: E402
from tests.helpers import smooth_u8  # noqa: E402

LR = (72, 104)
KW = dict(lr_level=3, device=0, denoising=False, denoise_rate=0.2, upscaler_model="realesrgan", batch_size=1, jit_mode=False, lr_hr_resize=False,
          model_name="RealESRGAN_x2plus", weights="synthetic", seed=3, lr_shape=LR, dtype="f16")

UNPICKLABLE = None   # the image server's shape of state: a module global the callback reads, set before start()


def main():
    global UNPICKLABLE
    assert not gpu_runtime_touched(), "this script must start clean"
    pipe = CallerPipeline(HipUpscalerService, **KW)
    pipe.lock = __import__("threading").Lock()      # (a spawned worker could not receive this object: locks do not pickle)
    UNPICKLABLE = {"marker": 41}
    orig = pipe.upscaler_on_queue

    def on_queue(entry):   # a closure over module state: nothing here survives pickling
        entry.profiler.set("marker", UNPICKLABLE["marker"] + 1)
        with pipe.lock:
            orig(entry)
    pipe.upscaler.on_queue = on_queue
    assert pipe.upscaler.start_method() == "fork"
    pipe.start()
    assert not gpu_runtime_touched(), "starting the worker must not initialise the GPU in the parent"
    frames = smooth_u8(77, (6, LR[0], LR[1], 3))
    dev = pipe.upscaler.device
    steps = ["a1b2", 1, "c3", 3, 4, "ff"]
    held = []
    for i, step in enumerate(steps):
        prof = CallerProfiler()
        prof.start("recoder.output")
        n = 1 if i != 3 else 2      # one two-frame job among the one-frame ones
        t = torch.tensor(frames[i:i + n], dtype=torch.uint8, device=dev)   # the parent touches the GPU only now, after the fork
        held.append(t)
        pipe.upscaler.push_job(CallerEntry(frames=t, audio_segment=None, step=step, elapsed=0, last_modified=0, profiler=prof), timeout=300)
    got = [pipe.sink.get_result(timeout=600) for _ in steps]
    assert [g["step"] for g in got] == steps, [g["step"] for g in got]
    pipe.stop()
    assert pipe.upscaler.start_method() == "spawn"   # from now on this process holds a HIP context: a later service would be spawned
    # the same frames through an in-process upscaler
    from sharkshark4k_amd import _capi, weights as W
    ctx = _capi.Context(0)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(W.rrdbnet_table(3, scale=2), W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, LR, None, False, False, None, 0.2)
    for i, g in enumerate(got):
        n = 1 if i != 3 else 2
        want = up(torch.from_numpy(frames[i:i + n]).cuda()).cpu()
        assert torch.equal(g["frames"], want), f"job {i}: frames differ"
        assert {"recoder.output", "upscaler.upscale", "fsrcnn.model", "upscaler.output", "marker"} <= set(g["keys"]), g["keys"]
    print("FORKED SERVICE OK", {"jobs": len(got), "worker": "forked", "parent_touched_gpu_before_start": False})


if __name__ == "__main__":
    main()
