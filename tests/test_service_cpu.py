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
        HipUpscalerService(jit_mode="trt")


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_hip_service_fails_loudly_without_gpu():
    svc = HipUpscalerService(denoising=False, upscaler_model="fsrcnn", scale=2)
    with pytest.raises(Exception):
        svc.proc_init()  # no silent CPU fallback
