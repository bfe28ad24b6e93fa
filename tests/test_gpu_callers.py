"""GPU: the drop-in boundary with the CALLERS' own objects: a six-field record and a set / start / end / data profiler
that are not this package's types (tests/caller_shapes.py) go through a real ``HipUpscalerService`` worker - spawned (this process
holds a HIP context) and forked (a fresh interpreter, the way the reference's callers start it) - and ``on_queue`` is a bound method
of a pipeline object that owns the service and the next one."""
import os
import subprocess
import sys

import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from tests.caller_shapes import CallerEntry, CallerPipeline, CallerProfiler
from tests.helpers import smooth_u8

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LR = (72, 104)
# the image server's constructor call, literally (image_pipeline.py:58-61), plus the knobs the reference hard-codes
KW = dict(lr_level=3, device=0, denoising=False, denoise_rate=0.2, upscaler_model="realesrgan", batch_size=1, jit_mode=False, lr_hr_resize=False,
          model_name="RealESRGAN_x2plus", weights="synthetic", seed=3, lr_shape=LR, dtype="f16")


@pytest.fixture(scope="module")
def want(ctx):
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(W.rrdbnet_table(3, scale=2), W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, LR, None, False, False, None, 0.2)
    frames = torch.from_numpy(smooth_u8(77, (8, LR[0], LR[1], 3)))
    return frames, torch.cat([up(frames[i:i + 1].cuda()).cpu() for i in range(8)])


def test_spawned_worker_takes_the_callers_record_and_profiler(want):
    frames, ref = want
    svc = HipUpscalerService(**KW)
    assert svc.start_method() == "spawn"      # this process holds a HIP context
    svc.start()
    try:
        dev = frames.cuda()
        steps = ["3f9a", 1, "00", 3, 4, 5, "zz", 7]      # the image server's step is a sha1 string (image_pipeline.py:283)
        for i, step in enumerate(steps):
            prof = CallerProfiler()
            prof.start("recoder.output")
            svc.push_job(CallerEntry(frames=dev[i:i + 1].clone(), audio_segment=None, step=step, elapsed=0, last_modified=0, profiler=prof), timeout=300)
        got = [svc.get_result(timeout=300) for _ in steps]
        assert [g.step for g in got] == steps
        for i, g in enumerate(got):
            assert type(g) is CallerEntry and type(g.profiler) is CallerProfiler
            assert torch.equal(g.frames.cpu(), ref[i:i + 1])
            assert {"recoder.output", "upscaler.upscale", "fsrcnn.model"} <= set(g.profiler.data) and "upscaler.output" in g.profiler.opened
            assert g.profiler.data["fsrcnn.model"] >= 0 and g.elapsed > 0
    finally:
        svc.stop()


def test_on_queue_bound_method_of_a_two_service_pipeline_spawned(want):
    frames, ref = want
    pipe = CallerPipeline(HipUpscalerService, **KW)
    pipe.start()
    try:
        dev = frames.cuda()
        for i in range(6):
            prof = CallerProfiler()
            prof.start("recoder.output")
            pipe.upscaler.push_job(CallerEntry(frames=dev[i:i + 1].clone(), audio_segment=torch.zeros(4), step=i, profiler=prof), timeout=300)
        got = [pipe.sink.get_result(timeout=300) for _ in range(6)]
        assert [g["step"] for g in got] == list(range(6))
        for i, g in enumerate(got):
            assert torch.equal(g["frames"], ref[i:i + 1])
            assert {"upscaler.upscale", "upscaler.output", "upscaler.output.queue", "fsrcnn.model"} <= set(g["keys"])
    finally:
        pipe.stop()


def test_forked_worker_from_an_untouched_parent_the_reference_callers_way():
    """tests/drive_forked_hip_service.py in a fresh interpreter: service built and started before the parent touches the GPU -> forked
    worker, unpicklable callback state, device tensors created in the parent afterwards; results byte for byte."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "drive_forked_hip_service.py")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FORKED SERVICE OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


class _KeepOnQueue:
    """An on_queue that does what the image server's does - take the result to the host and put it on the result queue - for a result
    that already IS on the host: ``.cpu()`` is free, ``.clone()`` a host memcpy (the clone is small here: a new big host tensor through
    a torch queue is exactly what ``host_results`` exists to avoid)."""

    def __init__(self, svc):
        self.q = svc.result_queue

    def __call__(self, entry):
        assert not entry.frames.is_cuda and entry.frames.is_pinned()
        self.q.put(type(entry)(frames=entry.frames.cpu().clone(), audio_segment=None, step=entry.step, elapsed=entry.elapsed,
                               last_modified=entry.last_modified, profiler=entry.profiler))


@pytest.mark.parametrize("with_on_queue", [False, True])
def test_host_results_arrive_as_host_tensors_bit_identical(want, with_on_queue):
    """``host_results``: the worker copies each result into its pinned result ring on its D2H stream; the consumer - ``on_queue`` inside the
    worker, or ``get_result()`` in the process that started the service - gets a uint8 HOST tensor (a view of the slot; through the queue
    it travels as a 100-byte handle).  More jobs than slots, one- and two-frame jobs, string steps."""
    frames, ref = want
    svc = HipUpscalerService(**dict(KW, batch_size=2))
    svc.host_results = True
    if with_on_queue:
        svc.on_queue = _KeepOnQueue(svc)
    svc.start()
    try:
        assert svc.result_ring is not None and svc.result_ring.slot_bytes >= 2 * 144 * 208 * 3
        dev = frames.cuda()
        jobs = [(i, i + 1) for i in range(8)] + [(0, 2), (2, 4), (4, 6), (6, 8)] + [(i, i + 1) for i in range(8)]     # 20 jobs through 8 slots
        for step, (a, b) in enumerate(jobs):
            svc.push_job(CallerEntry(frames=dev[a:b].clone(), audio_segment=None, step=f"s{step}", profiler=CallerProfiler()), timeout=300)
            if step % 4 == 3:        # consume as we go: a view is valid until HOST_RESULT_SLOTS later results
                for k in range(step - 3, step + 1):
                    g = svc.get_result(timeout=300)
                    a2, b2 = jobs[k]
                    assert g.step == f"s{k}" and isinstance(g.frames, torch.Tensor) and not g.frames.is_cuda
                    assert torch.equal(g.frames, ref[a2:b2]), f"job {k}"
    finally:
        svc.stop()
