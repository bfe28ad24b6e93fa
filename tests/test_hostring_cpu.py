"""CPU: the host-frame rings (``hostring.py``) and the dispatcher's slot accounting around them - the product's host-frame path with only
the device work of the worker replaced (``tests.test_node_cpu.CpuWorker`` is ``HipUpscalerService`` with three device-bound stages
overridden; its ``_host_job`` is the product's, in its no-GPU form: read the ring, run ``upscale``, write the ring)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd.hostring import HostFrames, HostRing, SlotPool
from sharkshark4k_amd.node import UpscalerNode
from tests.test_node_cpu import CpuWorker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child_writes(ring, slot, value, done):
    ring.view(slot, (4, 8)).fill_(value)
    done.set()


@pytest.mark.parametrize("method", ["fork", "spawn"])
def test_ring_is_one_memory_in_parent_and_child(method):
    ring = HostRing(3, 1000)
    assert ring.slot_bytes == 4096 and ring.nbytes == 3 * 4096 and ring.fits((4, 32, 32)) and not ring.fits((4097,))
    ring.view(1, (4, 8)).fill_(7)
    ctx = mp.get_context(method)
    done = ctx.Event()
    p = ctx.Process(target=_child_writes, args=(ring, 2, 9, done))
    p.start()
    assert done.wait(60)
    p.join(30)
    assert int(ring.view(2, (4, 8)).sum()) == 9 * 32 and int(ring.view(1, (4, 8)).sum()) == 7 * 32   # the child's bytes are here; ours stayed
    shape = ring.write(0, np.arange(24, dtype=np.uint8).reshape(1, 2, 4, 3))
    assert shape == (1, 2, 4, 3) and ring.view(0, shape).flatten().tolist() == list(range(24))
    ring.write(0, torch.full((2, 2, 2, 3), 5, dtype=torch.uint8))
    assert int(ring.view(0, (2, 2, 2, 3)).sum()) == 5 * 24
    with pytest.raises(ValueError):
        ring.view(3, (1,))
    with pytest.raises(ValueError):
        ring.view(0, (4097,))
    ring.close()


def test_slot_pool_and_descriptor():
    pool = SlotPool(2)
    a, b = pool.take(), pool.take()
    assert a is not None and b is not None and a != b and pool.take() is None
    pool.give_in(a[0])
    assert pool.take() is None          # an input slot alone is not a job's worth
    pool.give_out(a[1])
    assert pool.take() == a
    hf = HostFrames(slot=1, out_slot=0, shape=(4, 720, 1280, 3))
    assert len(hf) == 4 and not hf.result


def _node(tmp_path, n=2, **kw):
    return UpscalerNode(devices=list(range(n)), service_cls=CpuWorker, backend="gloo", upscaler_model="fsrcnn", scale=2, denoising=False, seed=5,
                        weights="synthetic", marker_dir=str(tmp_path), lr_shape=(4, 6), **kw)


def test_host_frames_go_through_the_rings_views_valid_until_next_poll(tmp_path):
    node = _node(tmp_path, 2, frame_skips=False, host_slots=4)
    assert node.services[0].host_rings[0].slot_bytes == 4096 and node.services[0].out_hw(4, 6) == (8, 12)
    node.start(timeout=300)
    try:
        frames = np.arange(16, dtype=np.uint8).reshape(16, 1, 1, 1).repeat(4, 1).repeat(6, 2).repeat(3, 3)   # numpy host frames, as a recorder has them
        steps = node.submit_batch(frames)                 # 4 jobs of 4 frames
        assert steps == [0, 1, 2, 3]
        got, deadline = [], time.monotonic() + 120
        views = []
        while len(got) < 4 and time.monotonic() < deadline:
            for e in node.poll(0.05):
                assert not e.frames.is_cuda and e.frames.shape == (4, 8, 12, 3)
                # a view of the worker's output ring, not a copy
                ring = node.services[e.step % 2].host_rings[1]
                assert ring._t.data_ptr() <= e.frames.data_ptr() < ring._t.data_ptr() + ring.nbytes
                got.append((e.step, e.frames[:, 1, 1, 0].tolist(), int(e.frames[0, 0, 0, 0])))
                views.append(e.frames)
        assert [g[0] for g in got] == [0, 1, 2, 3]
        for step, vals, rank in got:
            assert vals == list(range(4 * step, 4 * step + 4)) and rank == step % 2
        rep = node.report()
        assert rep["host_jobs"] == 4 and rep["host_fallback"] == 0 and rep["lost"] == 0 and rep["peer_copies"] == [0, 0]
        node.poll(0.0)
        for svc in node.services:      # every slot is back after the next poll()
            assert sorted(svc._slot_pool.free_in) == sorted(svc._slot_pool.free_out) == [0, 1, 2, 3]
        # a job bigger than a slot, and a tensor job, still work: the old way
        big = torch.zeros(4, 40, 40, 3, dtype=torch.uint8)
        steps = node.submit_batch(big)
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == steps and out[0].frames.shape == (4, 80, 80, 3) and node.report()["host_fallback"] == 1
    finally:
        node.stop()
        node.close()


def test_more_jobs_than_slots_no_skip_mode_finishes_skip_mode_drops(tmp_path):
    node = _node(tmp_path, 1, frame_skips=False, host_slots=2)
    node.start(timeout=300)
    try:
        frames = torch.arange(40, dtype=torch.uint8).view(40, 1, 1, 1).expand(40, 4, 6, 3).contiguous()
        steps = node.submit_batch(frames)                 # 10 jobs into 2 slots without a poll in between: finished results are moved out of the ring
        assert steps == list(range(10))
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == steps
        for e in out:
            assert e.frames[:, 1, 1, 0].tolist() == list(range(4 * e.step, 4 * e.step + 4))
        assert node.report()["lost"] == 0 and node.report()["dropped"] == 0
    finally:
        node.stop()
        node.close()
    node = _node(tmp_path, 1, frame_skips=True, host_slots=2)
    node.services[0].job_queue.put  # noqa: B018 - (queue exists; the worker is never started: nothing drains it)
    frames = torch.zeros(16, 4, 6, 3, dtype=torch.uint8)
    steps = node.dispatcher.submit_batch(frames)          # 4 jobs, 2 slots, nobody consumes: two are skipped like on a full queue
    assert steps == [0, 1] and node.report()["dropped"] == 2
    node.close()


def test_dispatcher_ceiling_with_eight_noop_workers():
    """tools/dispatcher_ceiling.py: 8 spawned no-op workers, 720p in / 1440p out through the rings - the parent's own ceiling must be
    above what 8 GPUs consume (8 x 130 frames/s at four-frame jobs, 8 x 125 at one-frame jobs)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dispatcher_ceiling.py"), "--workers", "8", "--seconds", "2", "--json"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = {row["job_frames"]: row for row in json.loads(r.stdout.strip().splitlines()[-1])}
    print(rows)
    # Measured on the idle 8-vCPU build container: 3 208 / 1 983 frames/s (profiles/NOTES_r06.md 2) against the 1 040 / 1 000 that 8 GPUs
    # consume.  This is a throughput figure of a shared CPU box (a busy neighbour has shown 1 266 / 867): the test FAILS only on a
    # dispatcher that is broken (an order of magnitude off), and WARNS when a run falls short of what 8 GPUs need.
    assert rows[4]["frames_per_s"] > 130 and rows[1]["frames_per_s"] > 125, rows
    if rows[4]["frames_per_s"] < 8 * 130 or rows[1]["frames_per_s"] < 8 * 125:
        import warnings
        warnings.warn(f"dispatcher ceiling below the 8-GPU demand on this (loaded?) box: {rows}")


def test_ring_view_travels_as_a_handle_and_unpickles_as_a_tensor():
    import pickle
    from sharkshark4k_amd.hostring import RingView
    ring = HostRing(2, 4096)
    ring.view(1, (2, 3)).copy_(torch.tensor([[1, 2, 3], [4, 5, 6]], dtype=torch.uint8))
    blob = pickle.dumps(RingView(ring, 1, (2, 3)))
    assert len(blob) < 300
    t = pickle.loads(blob)
    assert isinstance(t, torch.Tensor) and t.tolist() == [[1, 2, 3], [4, 5, 6]] and t.data_ptr() == ring.view(1, (2, 3)).data_ptr()
    ring.close()
    with pytest.raises(RuntimeError, match="has not mapped"):
        pickle.loads(blob)


def test_host_steps_inside_a_killed_worker_are_rescued_not_lost(tmp_path):
    """A worker dies with host jobs queued inside it: their frames are still in ITS input ring, which the parent holds - the dispatcher
    copies them into the living worker's ring and queues them again; every step arrives, in order, none is lost."""
    import signal
    node = _node(tmp_path, 2, frame_skips=False, host_slots=6, lost_after_s=30.0)
    node.start(timeout=300)
    try:
        frames = torch.arange(48, dtype=torch.uint8).view(48, 1, 1, 1).expand(48, 4, 6, 3).contiguous()
        # stop worker 1 so that its jobs pile up inside it, then kill it
        os.kill(node.services[1].proc.pid, signal.SIGSTOP)
        steps = node.submit_batch(frames)                 # 12 jobs: six for each worker
        assert steps == list(range(12))
        time.sleep(0.3)
        os.kill(node.services[1].proc.pid, signal.SIGKILL)
        node.services[1].proc.join(30)
        out = node.drain(steps, timeout=120)
        assert [e.step for e in out] == steps
        for e in out:
            assert e.frames[:, 1, 1, 0].tolist() == list(range(4 * e.step, 4 * e.step + 4))
            assert int(e.frames[0, 0, 0, 0]) == 0          # every job was computed by worker 0 in the end ... (rank stamp)
        rep = node.report()
        assert rep["lost"] == 0 and rep["rescued"] == 6 and rep["alive"] == [True, False], rep
    finally:
        node.stop()
        node.close()
