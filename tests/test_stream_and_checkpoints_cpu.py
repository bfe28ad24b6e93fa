"""CPU: stream-mode fan-out/fan-in and checkpoint key maps (§8 f4)."""
import queue

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import checkpoints as ck
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.stream import StreamDispatcher
from sharkshark4k_amd.util import Profiler
from tests.test_service_cpu import NearestDouble


class FakeService:
    """In-process stand-in with the queue attributes of BaseService (no worker process)."""

    def __init__(self, maxsize=2):
        self.job_queue = queue.Queue(maxsize=maxsize)
        self.result_queue = queue.Queue()

    def push_job_nowait(self, e):
        self.job_queue.put_nowait(e)

    def push_job(self, e, timeout=10):
        self.job_queue.put(e, timeout=timeout)

    def work(self, n=None):
        done = 0
        while not self.job_queue.empty() and (n is None or done < n):
            e = self.job_queue.get()
            e.profiler.set("upscaler.upscale", 0.004)
            e.frames = e.frames.repeat_interleave(2, 1).repeat_interleave(2, 2)
            self.result_queue.put(e)
            done += 1


def test_chunking_round_robin_and_ordered_fan_in():
    svcs = [FakeService(maxsize=8), FakeService(maxsize=8)]
    emitted = []
    d = StreamDispatcher(svcs, fps=24, on_result=lambda e: emitted.append(e.step))
    frames = torch.arange(24).view(24, 1, 1, 1).expand(24, 2, 2, 3).to(torch.uint8)
    steps = d.submit_batch(frames, audio_segment=np.zeros((44100, 2), np.float32), profiler=Profiler())
    assert d.small_batch_size == 4 and steps == list(range(6))
    assert svcs[0].job_queue.qsize() == 3 and svcs[1].job_queue.qsize() == 3  # step % G
    svcs[1].work()            # GPU 1 finishes first: nothing may be emitted before step 0 exists
    assert d.poll() == []
    svcs[0].work()
    out = d.drain(steps, timeout=5)
    assert [e.step for e in out] == list(range(6)) == emitted
    assert out[0].frames.shape == (4, 4, 4, 3) and int(out[5].frames[0, 0, 0, 0]) == 20
    assert out[0].audio_segment.shape[0] == 44100 // 6
    assert abs(out[0].profiler.data["upscaler.upscale.per_frame_ms"] - 1.0) < 1e-9


def test_frame_skip_backpressure():
    svc = FakeService(maxsize=2)
    d = StreamDispatcher([svc], fps=2)  # small_batch_size = min(4, fps) = 2
    frames = torch.zeros(10, 2, 2, 3, dtype=torch.uint8)
    steps = d.submit_batch(frames)
    assert d.small_batch_size == 2 and steps == [0, 1] and d.dropped == [2, 3, 4]
    svc.work()
    out = d.drain(steps, timeout=5)
    assert [e.step for e in out] == [0, 1]
    steps2 = d.submit_batch(frames[:2])  # stream continues after the dropped steps
    svc.work()
    assert [e.step for e in d.drain(steps2, timeout=5)] == [5]
    assert d.report()["dropped"] == 3


def test_dispatcher_with_real_worker_processes():
    svcs = [NearestDouble(), NearestDouble()]
    for s in svcs:
        s.start()
    try:
        d = StreamDispatcher(svcs, fps=24, frame_skips=False)
        frames = torch.arange(8).view(8, 1, 1, 1).expand(8, 2, 3, 3).to(torch.uint8).contiguous()
        steps = d.submit_batch(frames)
        out = d.drain(steps, timeout=120)
        assert [e.step for e in out] == [0, 1]
        assert out[1].frames.shape == (4, 4, 6, 3) and int(out[1].frames[0, 0, 0, 0]) == 4
    finally:
        for s in svcs:
            s.stop()


def test_fsrcnn_and_realesrgan_checkpoint_layouts():
    t = W.fsrcnn_table(3)
    ckpt = {"epoch": 1, "best_psnr": 0.0, "state_dict": {k: torch.from_numpy(v) for k, v in t.items()}}
    got = ck.fsrcnn_from_checkpoint(ckpt)
    assert list(got) == W.fsrcnn_keys() and all(np.array_equal(got[k], t[k]) for k in t)
    r = W.rrdbnet_table(1, num_block=1)
    got = ck.realesrgan_from_checkpoint({"params": {}, "params_ema": r}, "rrdbnet", num_block=1)
    assert list(got) == W.rrdbnet_keys(1)
    s = W.srvgg_table(1, num_conv=2)
    assert list(ck.realesrgan_from_checkpoint({"params": s}, "srvgg", num_conv=2)) == W.srvgg_keys(2)
    with pytest.raises(KeyError):
        ck.realesrgan_from_checkpoint({"params": {}}, "rrdbnet", num_block=1)


def test_bsvd_upstream_key_remap():
    """Build a checkpoint in the UPSTREAM layout (what bsvd/model.py:487-499 consumes) from our table
    by inverting the documented remaps, then check the importer recovers the table."""
    t = W.bsvd_table(5)
    up = {}
    for i, blk in enumerate(("temp1", "temp2")):
        for k, v in t.items():
            if not k.startswith(blk + "."):
                continue
            kk = k[len(blk) + 1:]
            for d in ("downc0.", "downc1."):
                if kk.startswith(d + "memconv."):
                    kk = d + "convblock.3." + kk[len(d + "memconv."):].replace("op.conv.", "net.")
            for u in ("upc2.", "upc1."):
                if kk.startswith(u + "memconv."):
                    kk = u + "convblock.0." + kk[len(u + "memconv."):].replace("op.conv.", "net.")
                elif kk.startswith(u + "convblock.0."):
                    kk = u + "convblock.1." + kk[len(u + "convblock.0."):]
            up[f"module.base_model.nets_list.{i}.{kk}"] = v
    got = ck.bsvd_from_checkpoint({"params": up})
    assert list(got) == W.bsvd_keys()
    assert all(np.array_equal(got[k], t[k]) for k in t)
