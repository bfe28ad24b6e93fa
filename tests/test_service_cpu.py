"""CPU: the service boundary (queues, worker process, entry type, profiler) behaves like the
reference's BaseService/BaseUpscalerService; the HIP service refuses to run without its extension."""
import pickle
import time
from queue import Empty

import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd.upscale.base_service import BaseService, ProcessDeadException  # noqa: F401
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService, LR_LEVELS
from sharkshark4k_amd.upscale.upscaler_base import BaseUpscalerService, UpscalerQueueEntry
from sharkshark4k_amd.util import Profiler, human_readable


class NearestDouble(BaseUpscalerService):
    """Test double standing in for the GPU service: x2 nearest on the CPU (NOT a product path)."""

    def proc_init(self):
        self.ready = True

    def upscale(self, frames):
        assert frames.ndim == 4 and frames.shape[-1] == 3
        return frames.repeat_interleave(2, 1).repeat_interleave(2, 2)


class NearestDoubleKw(NearestDouble):
    """The double with the constructor shape of the real service (``on_queue=`` keyword), for ``CallerPipeline``."""

    def __init__(self, on_queue=None, **_):
        super().__init__()
        self.on_queue = on_queue


class Boom(BaseUpscalerService):
    def proc_init(self):
        raise RuntimeError("init failed loudly")


def test_profiler_semantics():
    p = Profiler()
    assert p.end("never.started") == -1 and "never.started" not in p.data
    p.start("a"); time.sleep(0.01); e1 = p.end("a")
    p.start("a"); time.sleep(0.02); e2 = p.end("a")
    assert e1 > 0 and abs(p.data["a"] - (e1 + e2) / 2) < 1e-9  # running mean
    p.set("k", "v"); assert p.data["k"] == "v"
    assert human_readable(2048).endswith("KB")


def test_entry_and_service_pickle():
    e = UpscalerQueueEntry(frames=torch.zeros(1, 2, 2, 3, dtype=torch.uint8), step=3, profiler=Profiler())
    e2 = pickle.loads(pickle.dumps(e))
    assert e2.step == 3 and e2.frames.shape == (1, 2, 2, 3)
    svc = NearestDouble()
    state = svc.__getstate__()
    assert "proc" not in state and "job_queue" in state


def test_worker_roundtrip_and_stop():
    svc = NearestDouble()
    svc.start()
    try:
        for step in range(3):
            frames = torch.full((2, 4, 6, 3), step, dtype=torch.uint8)
            svc.push_job(UpscalerQueueEntry(frames=frames, step=step, profiler=Profiler()))
        got = [svc.get_result(timeout=60) for _ in range(3)]
        assert [g.step for g in got] == [0, 1, 2]
        assert got[2].frames.shape == (2, 8, 12, 3) and int(got[2].frames.max()) == 2
        assert got[0].elapsed >= 0 and "upscaler.upscale" in got[0].profiler.data
        assert got[0].profiler.is_open("upscaler.output")  # span left open for the consumer
        svc.wait_for_job_clear()
        with pytest.raises(Empty):
            svc.get_result(timeout=0.05)
    finally:
        svc.stop()
    assert not svc.proc.is_alive()


def test_on_queue_callback_runs_in_worker():
    svc = NearestDouble()
    svc.on_queue = _forward_to_result_queue(svc)
    svc.start()
    try:
        svc.push_job_nowait(UpscalerQueueEntry(frames=torch.zeros(1, 2, 2, 3, dtype=torch.uint8), step=9, profiler=Profiler()))
        r = svc.get_result(timeout=60)
        assert r.step == 109
    finally:
        svc.stop()


class _forward_to_result_queue:
    def __init__(self, svc):
        self.q = svc.result_queue

    def __call__(self, entry):
        entry.step += 100
        self.q.put(entry)


def test_worker_death_is_visible():
    svc = Boom()
    svc.start()
    assert svc.join(timeout=60) not in (0, None)  # exception escaped proc_main
    assert not svc.proc.is_alive()


def test_hip_service_configuration_mirrors_reference():
    svc = HipUpscalerService(lr_level=3, denoising=False, upscaler_model="realesrgan", batch_size=4)
    assert svc.lr_shape == (720, 1280) and svc.output_shape is None and svc.single_mode is False
    assert svc.scale == 4 and svc.jit_mode == "hip" and svc.hr_shape == (1440, 2560)
    assert HipUpscalerService(upscaler_model="fsrcnn", denoising=False).single_mode is True
    assert [HipUpscalerService(lr_level=i, denoising=False).lr_shape for i in range(6)] == LR_LEVELS
    with pytest.raises(Exception):
        HipUpscalerService(upscaler_model="egvsr")
    with pytest.raises(Exception):
        HipUpscalerService(jit_mode="tensorrt9")   # a name neither this build nor the reference's factories know
    # the callers' own literals: None (stream pipeline default) and False (the image server's "eager") select the one backend, and so do
    # the reference's backend names (a configuration written for it keeps working)
    for jm in (None, False, "hip", "trt", "t2trt", "jit", "ds"):
        assert HipUpscalerService(denoising=False, jit_mode=jm).jit_mode == "hip"
    image_server = HipUpscalerService(lr_level=3, device=0, denoising=False, denoise_rate=0.2, on_queue=None, upscaler_model="realesrgan",
                                      batch_size=1, jit_mode=False, lr_hr_resize=False)   # the image server's constructor call, literally
    assert image_server.single_mode is False and image_server.lr_hr_resize is False


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_hip_service_fails_loudly_without_gpu():
    from sharkshark4k_amd import _capi
    svc = HipUpscalerService(denoising=False, upscaler_model="fsrcnn", scale=2)
    with pytest.raises(Exception):
        svc.proc_init()  # no silent CPU fallback
    assert not _capi.GPU_TOUCHED   # no GPU, no runtime initialised: the process is as fork-safe as before


# ---------------------------------------------------------------------------------------------------------------------------------
# The callers' OWN objects: a six-field record and a set/start/end/data profiler that are not this package's types (tests/caller_shapes.py)

from tests.caller_shapes import CallerEntry, CallerPipeline, CallerProfiler  # noqa: E402


@pytest.mark.parametrize("method", ["fork", "spawn"])
def test_caller_shaped_entry_and_profiler_through_a_worker(method):
    """A worker fed the caller's record and profiler: nothing but the six fields and set / start / end / data is touched, the result
    is a record of the CALLER's type, a string step (the image server's sha1) passes through."""
    svc = NearestDouble()
    svc.mp_start_method = method
    svc.start()
    try:
        for step in (0, "9f2c01aa", 2):
            prof = CallerProfiler()
            prof.start("recoder.output")
            svc.push_job(CallerEntry(frames=torch.full((1, 4, 6, 3), 7, dtype=torch.uint8), audio_segment=torch.ones(5), step=step, profiler=prof))
        got = [svc.get_result(timeout=120) for _ in range(3)]
        assert [g.step for g in got] == [0, "9f2c01aa", 2]
        for g in got:
            assert type(g) is CallerEntry and type(g.profiler) is CallerProfiler
            assert g.frames.shape == (1, 8, 12, 3) and int(g.frames.min()) == 7 and g.audio_segment.shape == (5,)
            assert g.elapsed >= 0 and g.last_modified > 0
            assert {"recoder.output", "upscaler.upscale"} <= set(g.profiler.data) and "upscaler.output" in g.profiler.opened
    finally:
        svc.stop()
    assert svc.start_method() in ("fork", "spawn")


def test_start_method_follows_the_gpu_state(monkeypatch):
    from sharkshark4k_amd.upscale import base_service as bs
    svc = NearestDouble()
    monkeypatch.setattr(bs, "gpu_runtime_touched", lambda: False)
    assert svc.start_method() == "fork"            # an untouched parent does what the reference does
    monkeypatch.setattr(bs, "gpu_runtime_touched", lambda: True)
    assert svc.start_method() == "spawn"           # a parent that holds a HIP context must not fork
    svc.mp_start_method = "fork"
    with pytest.raises(RuntimeError, match="already initialised the GPU"):
        svc.start_method()
    svc.mp_start_method = "spawn"
    assert svc.start_method() == "spawn"


@pytest.mark.parametrize("method", ["fork", "spawn"])
def test_on_queue_as_bound_method_of_the_pipeline_that_owns_two_services(method):
    """The stream caller's wiring: ``on_queue`` is a bound method of an object that holds the upscaler and the NEXT service; it runs
    inside the upscaler's worker and pushes into the other service's queue from there.  Forked (the reference's way) nothing is
    pickled; spawned, the whole pipeline object travels into the worker with the service."""
    pipe = CallerPipeline(NearestDoubleKw)
    pipe.upscaler.mp_start_method = pipe.sink.mp_start_method = method
    pipe.start()
    try:
        for step in range(4):
            prof = CallerProfiler()
            prof.start("recoder.output")
            pipe.upscaler.push_job_nowait(CallerEntry(frames=torch.full((2, 3, 5, 3), step, dtype=torch.uint8), audio_segment=torch.zeros(3),
                                                      step=step, profiler=prof))
        got = [pipe.sink.get_result(timeout=120) for _ in range(4)]
        assert [g["step"] for g in got] == [0, 1, 2, 3]
        for g in got:
            assert g["shape"] == (2, 6, 10, 3) and g["sum"] == g["step"] * 2 * 6 * 10 * 3
            assert {"upscaler.upscale", "upscaler.output", "upscaler.output.queue", "upscaler.output.frames.shape"} <= set(g["keys"])
        assert pipe.forwarded == 0    # the callback ran in the worker's copy of the pipeline, not here
    finally:
        pipe.stop()


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="container-only: drives the reference's own caller modules")
def test_reference_callers_drive_the_service_with_one_import_swapped():
    """`tests/golden/drive_reference_callers.py`: the reference's real ``TwitchUpscalerPostStreamer`` (recorder callback -> OUR worker ->
    its ``upscaler_on_queue`` bound method -> its streamer's queue) and its real image-server module (its handler thread, its
    ``start_pipeline()`` constructor call, its ``pipeline_onqueue`` global) run against a CPU double of ``HipUpscalerService`` - with the
    reference's own ``UpscalerQueueEntry`` / ``Profiler`` / ``RecoderEntry`` / ``TwitchStreamerEntry`` objects in the queues."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "golden", "drive_reference_callers.py")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DRIVE OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


class HeldResults(NearestDouble):
    """A double whose results are 'on the device' for a while: ready 30 ms after the job was taken, up to two of them may be held."""

    def proc_init(self):
        self.ready_at = {}
        self.polls = 0

    def proc_job_recieved(self, job):
        entry = super().proc_job_recieved(job)
        self.ready_at[entry.step] = time.time() + (0.03 if entry.step == 0 else 1.0)
        return entry

    def proc_deliver_lag(self):
        return 2

    def proc_result_ready(self, entry):
        self.polls += 1
        return time.time() >= self.ready_at[entry.step]

    def proc_before_deliver(self, entry):
        entry.profiler.set("delivered_after_ready_s", time.time() - self.ready_at[entry.step])
        entry.profiler.set("polls", self.polls)


def test_held_results_leave_when_ready_not_when_a_successor_arrives():
    """The worker loop's held results (BaseService): an ISOLATED job's result leaves as soon as it is ready - no successor, no idle flush
    needed - in job order, and more than `lag` results are never held (the oldest then leaves at once, ready or not)."""
    svc = HeldResults()
    svc.start()
    try:
        t0 = time.time()
        svc.push_job(UpscalerQueueEntry(frames=torch.zeros(1, 2, 2, 3, dtype=torch.uint8), step=0, profiler=Profiler()))
        lone = svc.get_result(timeout=60)
        took = time.time() - t0
        assert lone.step == 0 and lone.profiler.data["polls"] >= 2            # it WAS held and polled ...
        assert 0 <= lone.profiler.data["delivered_after_ready_s"] < 0.5       # ... and left when it became ready (a poll interval is 0.2 ms; a loaded box gets slack)
        assert took < 5.0
        for step in range(1, 7):                                              # a burst: at most two are held, order is kept
            svc.push_job(UpscalerQueueEntry(frames=torch.zeros(1, 2, 2, 3, dtype=torch.uint8), step=step, profiler=Profiler()))
        got = [svc.get_result(timeout=60) for _ in range(6)]
        assert [g.step for g in got] == [1, 2, 3, 4, 5, 6]
        # (results of the burst are ready a second after their job: the six jobs are in long before that)
        forced = [g.step for g in got if g.profiler.data["delivered_after_ready_s"] < 0]   # pushed out by the lag bound before they were ready
        assert forced == [1, 2, 3, 4] and all(g.profiler.data["delivered_after_ready_s"] >= 0 for g in got[-2:])
    finally:
        svc.stop()
