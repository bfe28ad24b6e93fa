"""GPU: the node launcher (``node.UpscalerNode``) and the one-frame-job overlap with REAL ``HipUpscalerService`` workers.

A one-GPU box cannot show G > 1 GPUs, so:
* world 1 over ``nccl`` (RCCL) through the launcher with ``force_group``: the worker creates its communicator, rank 0 = itself loads,
  the 67 MB RRDBNet blob takes the device round trip through ``dist.broadcast``, the group is left, frames come back;
* two workers on ``cuda:0`` (the launcher picks ``gloo`` for workers that share a GPU: RCCL refuses duplicate devices): rank 0 loads, rank 1
  gets the blob, jobs alternate, results are bit-identical to an in-process upscaler built from the same table;
* one of the two is killed: the stream goes on over the survivor, ``report()`` says so, a replacement joins.
"""
import time

import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.node import UpscalerNode
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from tests.helpers import smooth_u8

pytestmark = pytest.mark.gpu
LR = (72, 104)
KW = dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=3, lr_shape=LR, dtype="f16")


@pytest.fixture(scope="module")
def want(ctx):
    """What every worker must return: the same network / service path built in-process from the same generated table."""
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(W.rrdbnet_table(3, scale=2), W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, LR, None, True, False, None, 1.0)
    frames = torch.from_numpy(smooth_u8(77, (12, LR[0], LR[1], 3)))
    one = torch.cat([up(frames[i:i + 1].cuda()).cpu() for i in range(12)])
    four = torch.cat([up(frames[i:i + 4].cuda()).cpu() for i in range(0, 12, 4)])
    assert torch.equal(one, four)   # a frame's bytes do not depend on the job it arrived in
    return frames, one


def test_node_world1_nccl_through_the_launcher(want):
    frames, ref = want
    node = UpscalerNode(devices=[0], force_group=True, fps=24, frame_skips=False, **KW)
    assert node.services[0].group.force and node.services[0].group.backend is None   # -> nccl in the worker
    with node:
        steps = node.submit_batch(frames.cuda())                       # 3 jobs of 4 frames
        out = node.drain(steps, timeout=300)
        assert [e.step for e in out] == [0, 1, 2]
        got = torch.cat([e.frames.cpu() for e in out])
        assert torch.equal(got, ref)
    assert node.alive() == [False]


def test_node_host_frames_through_the_pinned_rings(want):
    """Host frames are the node's native input: numpy frames -> the worker's pinned input ring -> H2D on the WORKER's copy stream -> the
    job -> D2H into the pinned output ring -> a host view at the sink.  Bit-identical to the resident path; 4-frame and 1-frame jobs
    (the latter alternate over the job sets with two more in flight), more jobs than ring slots."""
    frames, ref = want
    for fps, nslots in ((24, 3), (1, 6)):
        node = UpscalerNode(devices=[0], fps=fps, frame_skips=False, host_slots=nslots, **KW)
        with node:
            got = {}
            host = frames.numpy()
            for rep in range(3):     # 3 x 12 frames: 9 jobs of four (3 slots) / 36 jobs of one (6 slots)
                steps = node.submit_batch(host)
                deadline = time.monotonic() + 300
                while not set(steps) <= set(got) and time.monotonic() < deadline:
                    for e in node.poll(0.01):
                        assert not e.frames.is_cuda
                        got[e.step] = e.frames.clone()       # (a view of the ring: valid until the next poll)
            order = sorted(got)
            assert order == list(range(len(order))) and len(order) == (9 if fps == 24 else 36)
            allf = torch.cat([got[s] for s in order])
            assert torch.equal(allf, torch.cat([ref, ref, ref]))
            rep = node.report()
            assert rep["host_jobs"] == len(order) and rep["host_fallback"] == 0 and rep["lost"] == 0 and rep["dropped"] == 0 and rep["peer_copies"] == [0]
            # device tensors still pass through untouched, interleaved with host jobs
            steps = node.submit_batch(frames[:4].cuda()) + node.submit_batch(frames[4:8].numpy())
            out = node.drain(steps, timeout=300)
            assert torch.equal(torch.cat([e.frames.cpu() for e in out]), ref[:8])
        node.close()


def test_node_over_every_visible_gpu(want):
    """``devices=None``: one worker per visible GPU, RCCL (``nccl``) for the start-up broadcast when there is more than one - on a one-GPU box
    this is the world-1 case; on a multi-GPU node it is the first real N > 1 run of the product path (rank 0 loads, broadcast, fan-out
    ``step % G``, ordered fan-in, host rings per worker)."""
    frames, ref = want
    g = torch.cuda.device_count()
    node = UpscalerNode(devices=None, fps=24, frame_skips=False, **KW)
    assert node.devices == list(range(g)) and (node.backend is None)      # -> nccl in the workers when g > 1
    with node:
        host = frames.numpy()
        steps = []
        for _ in range(max(1, g)):
            steps += node.submit_batch(host)
        out = node.drain(steps, timeout=600)
        assert [e.step for e in out] == steps
        assert torch.equal(torch.cat([e.frames for e in out]), torch.cat([ref] * max(1, g)))
        rep = node.report()
        assert rep["lost"] == 0 and rep["host_jobs"] == len(steps) and rep["alive"] == [True] * g
        if g > 1:   # a device tensor on GPU 0 sent to every worker: the others copy it over once each and say so
            steps = node.submit_batch(frames.cuda(0))
            out = node.drain(steps, timeout=600)
            assert torch.equal(torch.cat([e.frames.cpu() for e in out]), ref)
            pc = node.report()["peer_copies"]
            assert pc[0] == 0 and all(c >= 1 for c in pc[1:min(g, 3)]), pc
    node.close()


def test_node_two_workers_one_gpu_rank0_loads_jobs_alternate(want):
    frames, ref = want
    node = UpscalerNode(devices=[0, 0], fps=1, frame_skips=False, **KW)   # fps 1 -> one-frame jobs (the image server's job size)
    assert node.backend == "gloo"
    with node:
        steps = node.submit_batch(frames.cuda())
        out = node.drain(steps, timeout=300)
        assert [e.step for e in out] == list(range(12))
        assert torch.equal(torch.cat([e.frames.cpu() for e in out]), ref)   # rank 1 runs rank 0's weights, bit for bit
        rep = node.report()
        assert rep["lost"] == 0 and rep["rerouted"] == 0 and rep["alive"] == [True, True]
        # worker 1 dies (a job it cannot parse; exit_on_error is off, so only that child goes)
        node.services[1].job_queue.put("not a job")
        deadline = time.monotonic() + 60
        while node.services[1].proc.is_alive() and time.monotonic() < deadline:
            time.sleep(0.05)
        assert node.alive() == [True, False]
        steps = node.submit_batch(frames[:6].cuda())
        out = node.drain(steps, timeout=300)
        assert [e.step for e in out] == steps == list(range(12, 18))
        assert torch.equal(torch.cat([e.frames.cpu() for e in out]), ref[:6])
        rep = node.report()
        assert rep["rerouted"] == 3 and rep["lost"] == 0 and rep["alive"] == [True, False]
        assert node.replace_dead(timeout=300) == [1]
        steps = node.submit_batch(frames[6:].cuda())
        out = node.drain(steps, timeout=300)
        assert torch.equal(torch.cat([e.frames.cpu() for e in out]), ref[6:]) and node.report()["alive"] == [True, True]


def test_one_frame_jobs_alternate_over_two_job_sets_bit_identical(want):
    """The worker's one-frame overlap, driven in-process: consecutive one-frame jobs run on three (context, model, stream) sets; a
    multi-frame job in between runs on set 0; every frame equals the single-set result; ``wait=False`` hands results over un-ordered
    (the worker loop orders them one job later, ``proc_before_deliver``), ``wait=True`` orders them on the current stream."""
    frames, ref = want
    svc = HipUpscalerService(device=0, **KW)
    svc.proc_init()
    assert svc.deliver_lag == 2 and len(svc._sets) == 1
    dev = frames.cuda()
    outs = [svc.upscale(dev[i:i + 1]) for i in range(5)]                 # ordered on the current stream: .cpu() below is safe
    assert len(svc._sets) == 3 and all(js["stream"] is not None for js in svc._sets)
    assert torch.equal(torch.cat(outs).cpu(), ref[:5])
    mixed = [svc.upscale(dev[0:1], wait=False), svc.upscale(dev[4:8], wait=False), svc.upscale(dev[1:2], wait=False), svc.upscale(dev[2:3], wait=False)]
    torch.cuda.synchronize()
    assert torch.equal(mixed[0].cpu(), ref[0:1]) and torch.equal(mixed[1].cpu(), ref[4:8])
    assert torch.equal(mixed[2].cpu(), ref[1:2]) and torch.equal(mixed[3].cpu(), ref[2:3])
    # switched off: one set, the current stream, as before
    svc1 = HipUpscalerService(device=0, overlap_jobs=False, **KW)
    svc1.proc_init()
    assert svc1.deliver_lag == 0
    assert torch.equal(torch.cat([svc1.upscale(dev[i:i + 1]) for i in range(3)]).cpu(), ref[:3]) and len(svc1._sets) == 1


def test_worker_loop_delivers_in_order_with_the_overlap(want):
    """The real worker process with deliver_lag = 2: twelve one-frame jobs pushed back to back, results in order and bit-identical; a
    lone job (nothing follows it) still comes back at once."""
    frames, ref = want
    from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
    from sharkshark4k_amd.util import Profiler
    svc = HipUpscalerService(device=0, **KW)
    svc.start()
    try:
        dev = frames.cuda()
        svc.push_job(UpscalerQueueEntry(frames=dev[0:1].clone(), step=100, profiler=Profiler()), timeout=300)
        lone = svc.get_result(timeout=300)
        assert lone.step == 100 and torch.equal(lone.frames.cpu(), ref[0:1])
        for i in range(12):
            svc.push_job(UpscalerQueueEntry(frames=dev[i:i + 1].clone(), step=i, profiler=Profiler()), timeout=60)
        got = [svc.get_result(timeout=300) for _ in range(12)]
        assert [g.step for g in got] == list(range(12))
        assert torch.equal(torch.cat([g.frames.cpu() for g in got]), ref)
        assert "fsrcnn.model" in got[3].profiler.data and "upscaler.upscale" in got[3].profiler.data
    finally:
        svc.stop()


def test_service_soak_mixed_job_sizes_through_one_worker():
    """`tools/service_soak.py`: 400 jobs of 1 / 2 / 4 frames, every input a fresh tensor that the producer drops right after the push, through one
    spawned worker with the job sets; every result byte for byte against an in-process single-set upscaler.  (Without the service's hold on a
    job's input until its end event has fired, ~ 2 jobs in 1000 come back wrong: `profiles/earlier/r05/r05_service_soak.txt`.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "service_soak.py"), "400"], cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SERVICE SOAK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_node_kill_rescue_and_auto_replace_with_host_frames(want):
    """Two real workers on cuda:0, host frames, ``auto_replace=True``: a worker is SIGKILLed with jobs inside it - those jobs are re-queued
    from its input ring (none lost), the stream goes on over the survivor, and poll() swaps a fresh worker in without blocking."""
    import os
    import signal
    frames, ref = want
    node = UpscalerNode(devices=[0, 0], fps=24, frame_skips=False, auto_replace=True, **KW)
    with node:
        host = frames.numpy()
        got = {}
        steps = node.submit_batch(host) + node.submit_batch(host)          # six jobs, three per worker
        os.kill(node.services[1].proc.pid, signal.SIGKILL)                 # exactly the child this node started
        deadline = time.monotonic() + 300
        while (not set(steps) <= set(got) or node.report()["replaced"] < 1) and time.monotonic() < deadline:
            for e in node.poll(0.02):
                got[e.step] = e.frames.clone()
        rep = node.report()
        assert sorted(got) == steps and rep["lost"] == 0 and rep["replaced"] == 1 and rep["alive"] == [True, True], rep
        assert torch.equal(torch.cat([got[s] for s in steps]), torch.cat([ref, ref]))
        out = node.drain(node.submit_batch(host), timeout=300)              # the replacement serves
        assert torch.equal(torch.cat([e.frames for e in out]), ref)
    node.close()
