#!/usr/bin/env python3
"""Drive the REFERENCE'S OWN CALLERS with one import swapped: is the service a drop-in for the objects they really build?

Container-only (needs /root/reference; never travels to the GPU box).  Nothing of the reference is copied: its modules are imported
and run.  Two flows, each with the reference's caller code untouched and ``FsrcnnUpscalerService`` replaced by a CPU double of THIS
package's ``HipUpscalerService`` (same class, same constructor, same ``BaseUpscalerService.proc_job_recieved`` / ``BaseService`` worker
loop; only the three device-bound stages - ``proc_init``'s model build, ``upscale`` and the device placement - are replaced by a
nearest x2 on the CPU, because this container has no GPU):

1. **stream pipeline** (``src/sharkshark/pipeline.py``): the real ``TwitchUpscalerPostStreamer`` is constructed (it builds the real
   ``TwitchRecoder`` and ``TwitchStreamer`` objects around the service), the upscaler worker is started, the real ``recoder_on_queue``
   is fed a real ``RecoderEntry`` + ``Profiler`` (it cuts jobs, builds the reference's ``UpscalerQueueEntry`` and pushes them), the
   real ``upscaler_on_queue`` runs INSIDE our worker as the bound method it is and pushes ``TwitchStreamerEntry`` records into the
   real streamer service's queue, where this script reads them; the real ``streamer_on_queue`` then reads the profiler keys.
2. **image server** (``src/sharkshark/image_server/image_pipeline.py``): the module is imported with its import of the service
   swapped; ITS handler thread calls ITS ``start_pipeline()`` (the literal constructor call with ``jit_mode=False``,
   ``exit_on_error=True``), requests are pushed the way ``upscale_image`` pushes them (sha1 string ``step``, per-request semaphore),
   ITS ``pipeline_onqueue`` runs inside our worker and reads the module global that only a forked child has.

Stubs, all third-party packages the image lacks: cv2, flask, basicsr, realesrgan, streamlink, av, redis / fastapi where imported.
``src/util/env_var.py`` is the user's credentials file the reference ships only as ``env_var.example.py``: the example is loaded
under that name, which is what a user does with it.

Usage:  python tests/golden/drive_reference_callers.py      (prints DRIVE OK / exits non-zero)
"""
from __future__ import annotations

import hashlib
import importlib
import importlib.util
import os
import sys
import threading
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService  # noqa: E402


class CpuDoubleOfHipService(HipUpscalerService):
    """``HipUpscalerService`` with the device-bound stages replaced (NOT a product path): constructor, attributes, worker loop,
    ``proc_job_recieved``, result delivery are the product's."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.ctor_seen = dict(kw)
        self.device = "cpu"   # (the callers create their frames on `upscaler.device`; this box has no GPU)

    def proc_init(self):
        self.torch_device = torch.device("cpu")
        self.deliver_lag = 0
        torch.set_num_threads(1)   # (a forked child must not enter the parent's OpenMP pool: this double computes with torch on the CPU)

    def upscale(self, frames, wait=True):
        assert isinstance(frames, torch.Tensor) and frames.ndim == 4 and frames.shape[-1] == 3 and frames.dtype == torch.uint8
        out = frames.repeat_interleave(2, 1).repeat_interleave(2, 2)
        if self.output_shape is not None:   # the pipeline overwrites this attribute (pipeline.py:46-50)
            out = out[:, :self.output_shape[0], :self.output_shape[1]]
        self.profiler.set("fsrcnn.model", 0.0)
        return out.contiguous()


class _Stub(types.ModuleType):
    """A missing third-party package: any attribute is a do-nothing class."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {"__init__": lambda self, *a, **kw: None})


def _stub(name):
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        n = ".".join(parts[:i])
        if n not in sys.modules:
            m = _Stub(n)
            m.__path__ = []
            sys.modules[n] = m


def _import_with_stubs(name, allowed=("cv2", "flask", "basicsr", "realesrgan", "streamlink", "av", "redis", "fastapi", "uvicorn", "requests_toolbelt")):
    stubbed = []
    for _ in range(64):
        try:
            return importlib.import_module(name), stubbed
        except ModuleNotFoundError as e:
            if e.name.split(".")[0] not in allowed:
                raise
            _stub(e.name)
            stubbed.append(e.name)
    raise RuntimeError("too many missing modules")


def _load_example_env():
    import src.util  # noqa: F401
    spec = importlib.util.spec_from_file_location("src.util.env_var", os.path.join(REF, "src", "util", "env_var.example.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    sys.modules["src.util.env_var"] = m


def nearest2(frames: np.ndarray) -> np.ndarray:
    return frames.repeat(2, 1).repeat(2, 2)


# --------------------------------------------------------------------------------------------------------------------------------
def drive_stream_pipeline():
    _load_example_env()
    P, stubbed = _import_with_stubs("src.sharkshark.pipeline")
    from src.stream.recoder import RecoderEntry
    from src.stream.streamer import TwitchStreamerEntry
    from src.upscale.upscaler_base import UpscalerQueueEntry as RefEntry
    from src.util.profiler import Profiler as RefProfiler
    P.FsrcnnUpscalerService = CpuDoubleOfHipService          # <- the one swapped import
    pipe = P.TwitchUpscalerPostStreamer(url="none", fps=8, device="cpu", lr_level=0, hr_level=0, denoising=False, frame_skips=False)
    svc = pipe.upscaler
    assert type(svc) is CpuDoubleOfHipService and svc.on_queue == pipe.upscaler_on_queue
    assert svc.ctor_seen["jit_mode"] is None and svc.ctor_seen["batch_size"] == 4 and svc.output_shape == (1440, 2560)
    assert svc.start_method() == "fork", "an untouched parent must fork, as the reference does: the caller's objects are not picklable"
    svc.output_shape = (90, 2560)   # (exercise the attribute override with a shape this toy frame size can show)
    svc.start()                      # only the upscaler: recorder / streamer need ffmpeg and the network
    try:
        rng = np.random.default_rng(0)
        frames = rng.integers(0, 256, (8, 48, 64, 3), dtype=np.uint8)
        audio = rng.random((44100, 2)).astype(np.float32)
        prof = RefProfiler()
        prof.start("recoder.output")
        pipe.recoder_on_queue(RecoderEntry(index=0, audio_segment=audio, frames=frames, fps=8, profiler=prof))   # the reference's own chunking + push
        got = [pipe.streamer.job_queue.get(timeout=60) for _ in range(2)]   # what the reference's upscaler_on_queue pushed, from inside OUR worker
        assert pipe.frame_step == 2
        want = nearest2(frames)[:, :90]
        for i, e in enumerate(got):
            assert type(e) is TwitchStreamerEntry and e.step == i and type(e.profiler) is RefProfiler
            assert np.array_equal(e.frames.numpy(), want[4 * i:4 * i + 4]), "frames differ"
            assert e.audio_segments.shape == (22050, 2)
            keys = set(e.profiler.data)
            assert {"recoder.output", "upscaler.upscale", "fsrcnn.model", "upscaler.output.queue", "recoder.output.entry"} <= keys, keys
            assert "upscaler.output" in e.profiler.start_ticks    # left open for the streamer to close (streamer.py:67)
            pipe.last_reported = 0
            pipe.streamer_on_queue(e)                              # the reference's reader of 'upscaler.upscale'
            assert e.profiler.data["upscaler.upscale.per_frame_ms"] >= 0
        # a raw entry of the reference's type straight into the queue comes back as an entry of the reference's type
        svc2 = CpuDoubleOfHipService(denoising=False, jit_mode=None)
        svc2.start()
        try:
            svc2.push_job(RefEntry(frames=torch.from_numpy(frames[:1]), audio_segment=None, step="abc", profiler=RefProfiler()))
            r = svc2.get_result(timeout=60)
            assert type(r) is RefEntry and r.step == "abc" and r.frames.shape == (1, 96, 128, 3) and r.elapsed >= 0
        finally:
            svc2.stop()
    finally:
        svc.stop()
    return {"flow": "stream pipeline", "stubbed": stubbed, "jobs": 2}


def drive_image_server():
    # the one swapped import: the module the image server takes the service (and the entry type) from
    from src.upscale import upscaler_base as ref_base
    swapped = types.ModuleType("src.upscale.fsrcnn_upscaler")
    swapped.FsrcnnUpscalerService = CpuDoubleOfHipService
    swapped.UpscalerQueueEntry = ref_base.UpscalerQueueEntry
    sys.modules["src.upscale.fsrcnn_upscaler"] = swapped
    import src.upscale
    src.upscale.fsrcnn_upscaler = swapped
    # flask: the blueprint's decorators must hand the functions back
    fl = types.ModuleType("flask")

    class Blueprint:
        def __init__(self, *a, **kw):
            pass

        def route(self, *a, **kw):
            return lambda f: f
    fl.Blueprint = Blueprint
    fl.Flask = type("Flask", (), {"__init__": lambda self, *a, **kw: None, "register_blueprint": lambda self, *a, **kw: None})
    sys.modules["flask"] = fl
    sys.path.append(os.path.join(REF, "src"))   # (the server's cache module imports `util` as a top-level package: it is run from src/)
    IP, stubbed = _import_with_stubs("src.sharkshark.image_server.image_pipeline")
    deadline = time.time() + 60
    while IP.upscaler is None and time.time() < deadline:   # ITS handler thread runs ITS start_pipeline()
        time.sleep(0.01)
    svc = IP.upscaler
    while not svc.proc.is_alive() and time.time() < deadline:   # (the module sets its global BEFORE it calls start(): wait for the worker itself)
        time.sleep(0.01)
    assert type(svc) is CpuDoubleOfHipService and svc.exit_on_error is True and svc.on_queue is IP.pipeline_onqueue
    assert svc.ctor_seen["jit_mode"] is False and svc.ctor_seen["lr_hr_resize"] is False and svc.ctor_seen["batch_size"] == 1
    assert svc.proc.is_alive()
    try:
        rng = np.random.default_rng(1)
        results = {}

        def request(i):   # what upscale_image does between decode and encode (image_pipeline.py:274-340), with the module's own tables
            img = rng.integers(0, 256, (40 + 2 * i, 56 + i, 3), dtype=np.uint8)
            my_id = hashlib.sha1(img.tobytes()).hexdigest()
            sema = threading.Semaphore(0)
            with IP.upscaler_queue_lock:
                IP.upscaler_queue_semas[my_id] = sema
            prof = IP.Profiler()
            prof.start("endpoint.proc")
            IP.upscaler.push_job(IP.UpscalerQueueEntry(frames=torch.tensor(img, dtype=torch.uint8, device=IP.upscaler.device).unsqueeze(0),
                                                       audio_segment=None, step=my_id, elapsed=0, last_modified=time.time(), profiler=prof), timeout=20)
            assert sema.acquire(timeout=60)
            with IP.upscaler_queue_lock:
                entry = IP.upscaler_queue_entries[my_id]
            assert entry.step == my_id and type(entry) is IP.UpscalerQueueEntry
            entry.profiler.end("endpoint.proc")
            results[i] = (img, entry)

        threads = [threading.Thread(target=request, args=(i,)) for i in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert len(results) == 6
        for i, (img, entry) in results.items():
            assert np.array_equal(entry.frames.numpy()[0], nearest2(img[None])[0])
            assert {"upscaler.upscale", "fsrcnn.model", "endpoint.proc"} <= set(entry.profiler.data)
    finally:
        svc.exit_on_error = False
        svc.stop()
    return {"flow": "image server", "stubbed": stubbed, "requests": 6}


if __name__ == "__main__":
    if not os.path.isdir(REF):
        print("no /root/reference here: nothing to drive")
        sys.exit(2)
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    out = []
    if which in ("both", "stream"):
        out.append(drive_stream_pipeline())
    if which in ("both", "image"):
        out.append(drive_image_server())
    for o in out:
        print(o)
    print("DRIVE OK")
