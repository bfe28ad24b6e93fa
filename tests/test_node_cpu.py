"""CPU: the node launcher (``sharkshark-4k_amd/node.py``) end to end over ``gloo`` - the product's multi-GPU path with only the
device-bound stages of the worker replaced.

``CpuWorker`` IS ``HipUpscalerService`` (same ``proc_init`` -> ``_node_group`` -> ``_shared_flat`` -> ``sharding.join_group`` /
``broadcast_weights`` / ``leave_group``, same ``BaseService`` worker loop, queues, ``ready_event``); it overrides the three stages
that need a GPU: ``_open_device`` (no HIP context), ``_build_models`` (keeps the blob it was handed instead of uploading it) and
``upscale`` (nearest x2 that stamps the worker's rank and a checksum of ITS weights into the frame).  So these tests show: G
spawned workers, ONLY rank 0 ran the weight loader, every worker holds bit-identical weights, jobs fan out ``step % G`` and come back
in order, a dead worker is routed around and counted, and a replacement joins the stream.
"""
import os
import time

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.node import UpscalerNode
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService


class CpuWorker(HipUpscalerService):
    """Test double for the device-bound stages only (NOT a product path)."""

    def __init__(self, marker_dir=None, **kw):
        self.marker_dir = marker_dir
        super().__init__(**kw)

    def _open_device(self):
        self.torch_device = torch.device("cpu")
        self._local_device = None

    def _build_models(self):
        from sharkshark4k_amd.upscale import model as factory
        def loader():
            # leaves a file behind: the test counts how many workers resolved the weights themselves
            open(os.path.join(self.marker_dir, f"loaded_by_rank{self.node_rank}_pid{os.getpid()}"), "w").close()
            if isinstance(self.weights, dict) and self.weights.get("sr") == "broken":
                raise FileNotFoundError("no such checkpoint (injected)")
            return factory.fsrcnn_flat(2, "synthetic", self.seed)
        self.flat = self._shared_flat("sr", factory.fsrcnn_desc(2), loader)
        self.model = self.denoise_model = None

    def _init_job_sets(self):
        self.deliver_lag = 0

    def upscale(self, frames, wait=True):
        assert frames.ndim == 4 and frames.shape[-1] == 3
        out = frames.repeat_interleave(2, 1).repeat_interleave(2, 2).clone()
        out[:, 0, 0, 0] = self.node_rank
        out[:, 0, 0, 1] = int(abs(float(self.flat.sum())) * 1000) % 251   # checksum of THIS worker's copy of the weights
        out[:, 0, 0, 2] = self.device
        return out


def _node(tmp_path, n=2, **kw):
    return UpscalerNode(devices=list(range(n)), service_cls=CpuWorker, backend="gloo", upscaler_model="fsrcnn", scale=2, denoising=False,
                        seed=5, weights=kw.pop("weights", "synthetic"), lost_after_s=kw.pop("lost_after_s", 5.0), marker_dir=str(tmp_path), **kw)


def _want_checksum(seed=5):
    return int(abs(float(W.flatten(W.fsrcnn_table(seed), W.fsrcnn_keys()).sum())) * 1000) % 251


def test_two_workers_rank0_loads_broadcast_fan_out_ordered(tmp_path):
    node = _node(tmp_path, 2, frame_skips=False)
    node.start(timeout=300)
    try:
        assert node.alive() == [True, True]
        # only rank 0 resolved the weights; the other worker got them through the group
        loaded = sorted(f.split("_pid")[0] for f in os.listdir(tmp_path))
        assert loaded == ["loaded_by_rank0"], loaded
        frames = torch.arange(24).view(24, 1, 1, 1).expand(24, 4, 6, 3).to(torch.uint8).contiguous()
        steps = node.submit_batch(frames)           # fps 24 -> jobs of 4 frames
        assert steps == list(range(6))
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == list(range(6))
        for e in out:
            assert e.frames.shape == (4, 8, 12, 3)
            assert int(e.frames[0, 0, 0, 0]) == e.step % 2 == int(e.frames[0, 0, 0, 2])   # job step ran on worker step % G
            assert int(e.frames[0, 0, 0, 1]) == _want_checksum()                            # ... which holds rank 0's weights, bit for bit
            assert int(e.frames[1, 1, 1, 0]) == e.step * 4 + 1
        rep = node.report()
        assert rep["lost"] == 0 and rep["dropped"] == 0 and rep["rerouted"] == 0 and rep["alive"] == [True, True]
    finally:
        codes = node.stop()
    assert node.alive() == [False, False] and len(codes) == 2


def test_kill_one_worker_stream_continues_and_loss_is_counted(tmp_path):
    node = _node(tmp_path, 2, frame_skips=False, lost_after_s=30.0)
    node.start(timeout=300)
    try:
        frames = torch.zeros(8, 4, 6, 3, dtype=torch.uint8)
        assert [e.step for e in node.drain(node.submit_batch(frames), timeout=120)] == [0, 1]
        # worker 1 dies with a job inside: exactly this PID, the one this test started
        node.services[1].job_queue.put("not a job: makes proc_job_recieved raise")   # the worker dies on it (exit_on_error is off)
        deadline = time.monotonic() + 60
        while node.services[1].proc.is_alive() and time.monotonic() < deadline:
            time.sleep(0.05)
        assert node.alive() == [True, False]
        steps = node.submit_batch(torch.zeros(16, 4, 6, 3, dtype=torch.uint8))   # steps 2..5: 3 and 5 would have gone to worker 1
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == steps == [2, 3, 4, 5]
        assert all(int(e.frames[0, 0, 0, 0]) == 0 for e in out)                   # all of them ran on the survivor
        rep = node.report()
        assert rep["rerouted"] == 2 and rep["lost"] == 0 and rep["alive"] == [True, False]
        # a step that was queued INSIDE the dead worker is declared lost without waiting lost_after_s (30 s here)
        dead = node.services[1]
        node.dispatcher._owner[node.dispatcher.frame_step] = dead                 # as if step 6 had been queued there before it died
        node.dispatcher.frame_step += 1
        t0 = time.monotonic()
        steps = node.submit_batch(torch.zeros(4, 4, 6, 3, dtype=torch.uint8))     # step 7
        assert [e.step for e in node.drain(steps, timeout=20)] == [7]
        assert time.monotonic() - t0 < 10 and node.report()["lost"] == 1
        # a replacement worker takes the slot (it loads the weights itself: the start-up group is gone) and the stream uses it
        assert node.replace_dead(timeout=300) == [1]
        assert node.alive() == [True, True]
        steps = node.submit_batch(torch.zeros(8, 4, 6, 3, dtype=torch.uint8))     # steps 8, 9
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == [8, 9]
        assert [int(e.frames[0, 0, 0, 2]) for e in out] == [0, 1] and int(out[1].frames[0, 0, 0, 1]) == _want_checksum()
    finally:
        node.stop()


def test_rank0_loader_failure_fails_every_worker_and_start_raises(tmp_path):
    node = _node(tmp_path, 2, weights={"sr": "broken"})
    with pytest.raises(RuntimeError, match="died during start-up"):
        node.start(timeout=300)
    for svc in node.services:
        svc.proc.join(timeout=60)
    assert node.alive() == [False, False]


def test_single_worker_node_needs_no_group(tmp_path):
    node = _node(tmp_path, 1, frame_skips=False)
    with node:
        out = node.drain(node.submit_batch(torch.zeros(4, 2, 2, 3, dtype=torch.uint8)), timeout=120)
        assert [e.step for e in out] == [0] and int(out[0].frames[0, 0, 0, 1]) == _want_checksum()


def test_auto_replace_swaps_a_fresh_worker_in_without_blocking_the_stream(tmp_path):
    """``auto_replace=True``: poll() notices the dead worker, starts a replacement in the background and swaps it in when it is ready;
    meanwhile the stream runs over the survivor."""
    import signal
    node = _node(tmp_path, 2, frame_skips=False, auto_replace=True)
    node.start(timeout=300)
    try:
        os.kill(node.services[1].proc.pid, signal.SIGKILL)
        node.services[1].proc.join(30)
        frames = torch.arange(8).view(8, 1, 1, 1).expand(8, 4, 6, 3).to(torch.uint8).contiguous()
        seen, deadline = [], time.monotonic() + 240
        while node.report()["replaced"] < 1 and time.monotonic() < deadline:
            steps = node.submit_batch(frames)             # two jobs per round: the stream does not wait for the replacement
            t_end = time.monotonic() + 30
            while len(seen) < steps[-1] + 1 and time.monotonic() < t_end:
                seen += [e.step for e in node.poll(0.05)]
        rep = node.report()
        assert rep["replaced"] == 1 and rep["alive"] == [True, True] and rep["replacing"] == [] and rep["lost"] == 0, rep
        assert seen == list(range(len(seen))) and len(seen) >= 2
        steps = node.submit_batch(frames)                 # both workers serve again
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == steps
        assert {int(e.frames[0, 0, 0, 2]) for e in out} == {0, 1}      # (the double stamps its device: the replacement in slot 1 takes jobs again)
    finally:
        node.stop()
        node.close()
