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


def test_lost_result_does_not_stall_the_stream():
    """A queued job whose result never comes back (dead worker / full result queue) is declared lost
    after lost_after_s, counted, and the later steps are emitted; dropped steps are pruned."""
    svcs = [FakeService(maxsize=8), FakeService(maxsize=8)]
    d = StreamDispatcher(svcs, fps=24, lost_after_s=0.2)
    steps = d.submit_batch(torch.zeros(16, 2, 2, 3, dtype=torch.uint8))
    assert steps == [0, 1, 2, 3]
    svcs[0].job_queue.get()      # step 0 vanishes inside service 0
    svcs[0].work(); svcs[1].work()
    assert d.poll() == []        # steps 1-3 are held back for step 0 ...
    out = d.drain([1, 2, 3], timeout=5)
    assert [e.step for e in out] == [1, 2, 3]   # ... but only for lost_after_s
    rep = d.report()
    assert rep["lost"] == 1 and rep["dropped"] == 0 and rep["pending"] == 0 and d.dropped == []


def test_late_result_is_discarded_and_order_never_rewinds():
    """A result that arrives AFTER its step was declared lost (a slow first job while the other service keeps
    returning) must not be emitted behind newer frames, must not move next_emit backwards, and must not make the
    following batch stall or be counted as lost again."""
    svcs = [FakeService(maxsize=8), FakeService(maxsize=8)]
    emitted = []
    d = StreamDispatcher(svcs, fps=24, lost_after_s=0.2, on_result=lambda e: emitted.append(e.step))
    assert d.submit_batch(torch.zeros(16, 2, 2, 3, dtype=torch.uint8)) == [0, 1, 2, 3]
    slow = svcs[0].job_queue.get()               # step 0 is still being worked on ...
    svcs[0].work(); svcs[1].work()
    assert [e.step for e in d.drain([1, 2, 3], timeout=5)] == [1, 2, 3]
    assert d.next_emit == 4
    slow.profiler.set("upscaler.upscale", 0.004)
    svcs[0].result_queue.put(slow)               # ... and comes back after the stream has moved on
    assert d.poll() == [] and d.next_emit == 4   # discarded, nothing rewinds
    steps = d.submit_batch(torch.zeros(8, 2, 2, 3, dtype=torch.uint8))
    svcs[0].work(); svcs[1].work()
    assert [e.step for e in d.drain(steps, timeout=1)] == [4, 5]   # no second stall
    assert emitted == [1, 2, 3, 4, 5]
    rep = d.report()
    assert rep["lost"] == 1 and rep["late"] == 1 and rep["pending"] == 0


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


# ------------------------------------------------------------------------------ weight sources of the factories / service
def test_factories_never_fall_back_to_made_up_weights(tmp_path, monkeypatch):
    """without weights the factories look for the reference's checkpoint files and
    raise when they are missing; 'synthetic' is an explicit opt-in; paths, torch.load dicts and
    state-dict tables all resolve to the same table."""
    from sharkshark4k_amd.upscale import model as factory
    monkeypatch.delenv("SS4K_CHECKPOINT_DIR", raising=False)
    for fn in (lambda: factory.fsrcnn_table_from(None, 4), lambda: factory.esrgan_table_from("RealESRGAN_x2plus"),
               lambda: factory.esrgan_table_from("realesr-general-x4v3"), lambda: factory.bsvd_table_from(None)):
        with pytest.raises(FileNotFoundError) as ei:
            fn()
        assert "synthetic" in str(ei.value) and ".pth" in str(ei.value)
    # the reference's file names inside checkpoint_dir (fsrcnn/factory.py:8-10, realesrgan/factory.py:140-157, bsvd/factory.py:35)
    t = W.fsrcnn_table(3)
    torch.save({"epoch": 3, "state_dict": {k: torch.from_numpy(v) for k, v in t.items()}}, tmp_path / "fsrcnn_x2-T91.pth")
    got = factory.fsrcnn_table_from(None, 2, checkpoint_dir=str(tmp_path))
    assert all(np.array_equal(got[k], t[k]) for k in t)
    monkeypatch.setenv("SS4K_CHECKPOINT_DIR", str(tmp_path))
    assert np.array_equal(factory.fsrcnn_table_from(None, 2)["deconv.weight"], t["deconv.weight"])
    with pytest.raises(FileNotFoundError):
        factory.fsrcnn_table_from(None, 4)  # only the x2 file is there
    # path / dict / table for one model are the same weights
    same = [factory.fsrcnn_table_from(spec, 2) for spec in (str(tmp_path / "fsrcnn_x2-T91.pth"), {"state_dict": t}, t)]
    assert all(np.array_equal(s_["shrink.0.bias"], t["shrink.0.bias"]) for s_ in same)
    assert np.array_equal(factory.fsrcnn_table_from("synthetic", 2, seed=3)["shrink.0.bias"], t["shrink.0.bias"])


def test_dni_blend_of_two_checkpoints_and_bsvd_remap_through_the_factory(tmp_path):
    from sharkshark4k_amd.upscale import model as factory
    a, b = W.srvgg_table(1), W.srvgg_table(2)
    torch.save({"params": {k: torch.from_numpy(v) for k, v in a.items()}}, tmp_path / "realesr-general-x4v3.pth")
    torch.save({"params": {k: torch.from_numpy(v) for k, v in b.items()}}, tmp_path / "realesr-general-wdn-x4v3.pth")
    arch, kw, t = factory.esrgan_table_from("realesr-general-x4v3", denoise_rate=0.25, checkpoint_dir=str(tmp_path))
    k = "body.4.weight"
    assert arch == "srvgg" and np.allclose(t[k], 0.25 * a[k] + 0.75 * b[k])   # realesrgan/factory.py:152-157
    _, _, t1 = factory.esrgan_table_from("realesr-general-x4v3", denoise_rate=1, checkpoint_dir=str(tmp_path))
    assert np.array_equal(t1[k], a[k])                                          # no blend at strength 1
    with pytest.raises(FileNotFoundError):  # a checkpoint without its wdn partner cannot be blended
        factory.esrgan_table_from("realesr-general-x4v3", 0.5, weights={"params": a})
    _, _, t2 = factory.esrgan_table_from("realesr-general-x4v3", 0.5, weights={"params": a}, weights_wdn={"params": b})
    assert np.allclose(t2[k], 0.5 * a[k] + 0.5 * b[k])
    # BSVD in the UPSTREAM key layout goes through the nets_list / convblock->memconv remap
    tb = W.bsvd_table(5)
    up = {}
    for i, blk in enumerate(("temp1", "temp2")):
        for kk, v in tb.items():
            if kk.startswith(blk + "."):
                r = kk[len(blk) + 1:]
                for d in ("downc0.", "downc1."):
                    if r.startswith(d + "memconv."):
                        r = d + "convblock.3." + r[len(d + "memconv."):].replace("op.conv.", "net.")
                for u in ("upc2.", "upc1."):
                    if r.startswith(u + "memconv."):
                        r = u + "convblock.0." + r[len(u + "memconv."):].replace("op.conv.", "net.")
                    elif r.startswith(u + "convblock.0."):
                        r = u + "convblock.1." + r[len(u + "convblock.0."):]
                up[f"base_model.nets_list.{i}.{r}"] = torch.from_numpy(v)
    torch.save({"params": up}, tmp_path / "bsvd-32.pth")
    got = factory.bsvd_table_from(None, checkpoint_dir=str(tmp_path))
    assert list(got) == W.bsvd_keys() and all(np.array_equal(got[q], tb[q]) for q in tb)


def test_service_weight_argument_is_validated():
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    assert HipUpscalerService(denoising=False).weights is None                 # -> checkpoint_dir lookup in proc_init
    assert HipUpscalerService(denoising=False, weights="synthetic").weights == "synthetic"
    with pytest.raises(TypeError):
        HipUpscalerService(denoising=False, weights="random")


class _NotATensor:   # module-level so that pickle can name it
    def __init__(self):
        self.x = 1


def test_checkpoint_files_load_tensors_only_with_a_named_opt_in(tmp_path, monkeypatch):
    """Checkpoint FILES are unpickled tensors-only (a .pth found in a directory is not trusted code).  A file with the usual
    training wrappers ({'state_dict': ..., 'epoch': ...}) loads; one that pickles an arbitrary object is refused with a message
    that names the explicit opt-in, SS4K_UNSAFE_TORCH_LOAD=1 - with which it loads as the reference's torch.load would."""
    import torch
    from sharkshark4k_amd.upscale import model as factory
    table = W.fsrcnn_table(3)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in table.items()}   # the table is in the reference's key names
    good = tmp_path / "wrapped.pth"
    torch.save({"state_dict": sd, "epoch": 12, "meta": {"lr": 1e-3, "name": "T91"}}, good)
    monkeypatch.delenv("SS4K_UNSAFE_TORCH_LOAD", raising=False)
    kind, obj = factory._load_checkpoint(str(good), "FSRCNN", "x.pth", None)
    assert kind == "ckpt" and set(obj["state_dict"]) == set(sd) and obj["epoch"] == 12
    bad = tmp_path / "objects.pth"
    torch.save({"state_dict": sd, "hook": _NotATensor()}, bad)
    with pytest.raises(RuntimeError, match="SS4K_UNSAFE_TORCH_LOAD=1"):
        factory._load_checkpoint(str(bad), "FSRCNN", "x.pth", None)
    monkeypatch.setenv("SS4K_UNSAFE_TORCH_LOAD", "1")
    kind, obj = factory._load_checkpoint(str(bad), "FSRCNN", "x.pth", None)
    assert kind == "ckpt" and isinstance(obj["hook"], _NotATensor)
