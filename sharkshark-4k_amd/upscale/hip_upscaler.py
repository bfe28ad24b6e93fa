"""``HipUpscalerService``: drop-in for the reference's ``FsrcnnUpscalerService``.

Same constructor arguments, attributes (``lr_shape``, ``output_shape``, ``on_queue``,
``exit_on_error``, ``device``, ``single_mode`` ...) and ``upscale(frames u8 NHWC) -> u8 NHWC``
contract as ``src/upscale/fsrcnn_upscaler.py:86-326``; all arithmetic runs in libss4k_hip.so
(``jit_mode='hip'``, the one backend of this build).  Extra keyword arguments that the reference
hard-codes are exposed: ``scale`` (FSRCNN factor, reference: always 4, fsrcnn_upscaler.py:101),
``model_name`` (reference: ``ArgsData.model_name``, realesrgan/factory.py:88), ``dtype`` and an
``lr_shape`` override.

Weights.  The reference always loads real checkpoints (``fsrcnn_x4-T91.pth``, the downloaded
RealESRGAN ``.pth`` files with the DNI blend, ``bsvd-32.pth``); so does this service:

* ``weights=None`` (default): the reference's own file names are looked up in ``checkpoint_dir``
  (else ``$SS4K_CHECKPOINT_DIR``); ``proc_init`` raises ``FileNotFoundError`` when one is missing.
* ``weights={'sr': X, 'sr_wdn': X, 'denoise': X}`` with X a ``.pth`` path, the dict ``torch.load``
  returns (``state_dict`` / ``params_ema`` / ``params``; BSVD's ``nets_list`` remap is applied) or a
  state-dict table; missing entries fall back to the ``checkpoint_dir`` lookup.
* ``weights='synthetic'``: deterministic generated weights for every model - an explicit opt-in
  used by the tests and ``bench.py`` only (logged loudly: the frames are noise).

Multi-GPU (``group=sharding.GroupSpec(rank, world, ...)``, set by ``node.UpscalerNode``): the worker joins the node's process group
in ``proc_init`` (RCCL: backend ``nccl``), ONLY RANK 0 resolves ``weights`` (reads / blends / generates the state dicts) and every
blob reaches the other workers through one ``sharding.broadcast_weights`` each; then the group is left and the workers are
independent.  The reference builds one service on ``device=0`` (``src/sharkshark/pipeline.py:20,41-50``); SURVEY.md 8(e).

One-frame jobs (``overlap_jobs=True``, batched path only - the image server's caller, ``image_pipeline.py:54-64,280-287``): consecutive
one-frame jobs alternate over ``overlap_sets`` (3) job sets (context + model + upscaler + stream), so that job i + 1's launches fill the
launch boundaries and partly filled tile rounds of job i - what frame lanes do inside a multi-frame job.  Frames are bit-identical to the
single-set path (same kernels, same weights).  The worker keeps at most ``overlap_sets - 1`` results of ALTERNATING jobs back while it
enqueues their successors, and hands each over as soon as its end event has fired (``BaseService``: held results; a multi-frame job flushes
them and leaves at once, as in the reference's loop); ``upscale()`` called directly stays synchronous with the current stream unless ``wait=False``.
Measured on one box (RRDBNet x2, 720p, ``profiles/earlier/r05/r05_n1_probe_sets.txt``): one set 108.8 frames/s, two 122.8, three 125.6, four 124.9 -
against 127.5 for four-frame jobs; two- and four-frame jobs gain nothing from alternating (125.5 / 125.7 against 125.0 / 127.5), so they
stay on set 0.  Cost per extra set: a copy of the SR weights and activation workspace (RRDBNet x2 at 720p: 67 MB + 1.25 GB), built on the
first one-frame job.  HIP serves a process's streams from a few hardware queues: the sets' streams are checked once to really run side by
side (``_check_streams``).

Host frames (``host_rings``, set by ``node.UpscalerNode``; ``hostring.py``): a job whose ``frames`` is a ``HostFrames`` descriptor names a slot
of this worker's pinned input ring; the worker copies it to the device on its own copy stream, runs the job, copies the result into the
named slot of the pinned output ring on a second copy stream and answers with a descriptor.  One more job is enqueued before a result is
waited for (two jobs in flight), so both copies run under the neighbouring jobs' kernels (``pipeline.py:84-93`` / ``streamer.py:92-98`` are the
reference's host <-> device hops).  A device tensor that lives on ANOTHER GPU is copied over once, counted in ``'upscaler.input.peer_copies'``.
"""
from __future__ import annotations

import collections
import contextlib
import sys
from typing import Mapping, Optional

import torch

from .. import sharding
from ..hostring import HostFrames
from ..util.profiler import Profiler
from .upscaler_base import BaseUpscalerService, UpscalerQueueEntry, answer, record_span  # noqa: F401

LR_LEVELS = [(360, 640), (540, 960), (630, 1120), (720, 1280), (900, 1600), (1080, 1920)]


def under_profiler() -> bool:
    """A profiler has loaded itself into this process (rocprofv3 sets ROCP_TOOL_LIBRARIES and preloads librocprofiler-sdk): counter
    collection serialises kernels, so every "do these two streams run side by side?" measurement fails - the library skips its lane check
    then (models.cpp ss4k_ctx::lane_check) and the service its stream vetting (eight candidates x ~ 600 probe launches per stream otherwise)."""
    import os
    env = os.environ
    return bool(env.get("ROCP_TOOL_LIBRARIES")) or any(t in env.get("LD_PRELOAD", "") for t in ("rocprofiler", "roctracer"))


def log(*args, **kwargs):
    # stderr: a process that prints a result on stdout (bench.py's one JSON line) may run a service in-process
    kwargs.setdefault("file", sys.stderr)
    print(f"HipUpscalerService: {' '.join(str(a) for a in args)}", **kwargs)


#: the reference's own backend names (realesrgan/factory.py:175-230, fsrcnn/factory.py:17-69, bsvd/factory.py:21-75)
REFERENCE_BACKENDS = ("trt", "t2trt", "jit", "ds", "trt_vol")

class HipUpscalerService(BaseUpscalerService):
    profiler: Profiler
    #: (input ring, output ring) of pinned shared host memory for ``HostFrames`` jobs (hostring.py); None: every job carries a tensor
    host_rings = None
    #: True (set before ``start()``): a worker's results are HOST tensors instead of device tensors - copied by the worker on its D2H stream
    #: into a ring of ``HOST_RESULT_SLOTS`` pinned shared-memory slots (``result_ring``, created by ``start()`` in the calling process).  For
    #: consumers that want the frames on the host anyway (the image server: image_pipeline.py:38-47,123): ``on_queue`` gets a tensor view of
    #: the slot, ``get_result()`` / ``result_queue.get()`` in the process that started the service (or a child of it) get the same view - it
    #: travels as ~ 100 bytes, not as a new 11 MB shared-memory segment per result (~ 90 ms each in a process that holds a HIP context,
    #: profiles/r06_latency.txt).  A view stays valid until ``HOST_RESULT_SLOTS`` later results; ``.clone()`` to keep.  A result bigger than a
    #: slot (``host_result_bytes``; default: ``batch_size`` frames of ``out_hw(lr_shape)``) stays a device tensor.
    host_results = False
    host_result_bytes = None
    result_ring = None
    HOST_RESULT_SLOTS = 8

    def __init__(self, lr_level=3, device=0, on_queue=None, denoising=True, denoise_rate=1.0,
                 upscaler_model="realesrgan", batch_size=1, jit_mode="hip", lr_hr_resize=True,
                 # knobs the reference hard-codes
                 scale=4, model_name=None, dtype="f16", weights=None, checkpoint_dir: Optional[str] = None,
                 lr_shape=None, single_mode=None, seed=0, model_flags=0, fsrcnn_dtype="f32",
                 group: Optional[sharding.GroupSpec] = None, overlap_jobs=True, overlap_sets=3, overlap_max_frames=1):
        # None (the stream pipeline's default: "the compiled backend", pipeline.py:23,41-44) and False (the image server: "eager",
        # image_pipeline.py:58-61) both mean "whatever this build runs the network with" to a caller that cannot know about 'hip';
        # so does a backend of the REFERENCE asked for by name ('trt', 't2trt', 'jit', 'ds': realesrgan/factory.py:175-230, fsrcnn/factory.py:17-69 -
        # a configuration written for the reference keeps working; said once on stderr).  Any other name is refused, as the reference's
        # factories refuse a name they do not know.
        if jit_mode in REFERENCE_BACKENDS:
            print(f"HipUpscalerService: jit_mode={jit_mode!r} names a backend of the reference; this build runs every network on its one "
                  f"backend, 'hip'", file=sys.stderr)
        elif not (jit_mode is None or jit_mode is False or jit_mode == "hip"):
            raise Exception(f"jit_mode={jit_mode!r}: this build has one backend, 'hip' (None, False and the reference's names select it too)")
        if upscaler_model not in ("fsrcnn", "realesrgan"):
            raise Exception(upscaler_model)
        self.lr_shape = tuple(lr_shape) if lr_shape is not None else LR_LEVELS[lr_level]
        self.scale = scale
        self.denoise_rate = denoise_rate
        self.hr_shape = (1440, 2560)  # set and never read by the reference either (fsrcnn_upscaler.py:104)
        self.device = device
        self.on_queue = on_queue
        self.output_shape = None
        self.upscaler_model = upscaler_model
        self.single_mode = (upscaler_model != "realesrgan") if single_mode is None else bool(single_mode)
        self.denoising = denoising
        self.batch_size = batch_size
        self.jit_mode = "hip"
        self.lr_hr_resize = lr_hr_resize
        self.model_name = model_name
        self.dtype = dtype
        if not (weights is None or weights == "synthetic" or isinstance(weights, Mapping)):
            raise TypeError("weights must be None (checkpoint_dir lookup), 'synthetic' or {'sr'|'sr_wdn'|'denoise': path | checkpoint | table}")
        self.weights = weights if isinstance(weights, str) or weights is None else dict(weights)
        self.checkpoint_dir = checkpoint_dir
        self.seed = seed
        self.fsrcnn_dtype = fsrcnn_dtype  # 'f32': fp32-accurate (parity bar); 'f16': the reference engine's precision, ~2x the rate
        self.model_flags = int(model_flags)  # SS4K_MODEL_* routing switches for the SR model (include/ss4k.h)
        self.group = group
        self.overlap_jobs = bool(overlap_jobs)
        self.overlap_sets = max(2, int(overlap_sets))            # job sets that consecutive small jobs alternate over
        self.overlap_max_frames = int(overlap_max_frames)        # jobs of up to this many frames alternate (bigger ones overlap with themselves: frame lanes)
        super().__init__()

    def net_scale(self) -> int:
        """The SR network's own factor (no GPU needed: from the model name / ``scale``)."""
        from . import model as factory
        if self.upscaler_model == "fsrcnn":
            return int(self.scale)
        kind, kw = factory.REALESRGAN_ZOO[self.model_name or factory.DEFAULT_REALESRGAN]
        return int(kw["scale"] if kind == "rrdbnet" else kw["upscale"])

    def out_hw(self, h: int, w: int):
        """(H, W) of the frames ``upscale`` returns for ``h x w`` input frames - the rule of ``ss4k_upscaler_out_shape`` (csrc/api.cpp
        ``Upscaler::out_shape``; reference: fsrcnn_upscaler.py:174-176,223-233,236-241,316-326), evaluated on the host so that a launcher
        without a HIP context can size result buffers."""
        lh, lw = h, w
        if self.single_mode or ((w > self.lr_shape[1] or h > self.lr_shape[0]) and self.lr_hr_resize):
            lh, lw = self.lr_shape
        if self.output_shape is not None and (self.single_mode or self.lr_hr_resize):
            return int(self.output_shape[0]), int(self.output_shape[1])
        k = self.net_scale()
        return lh * k, lw * k

    # worker side -----------------------------------------------------------------------------
    def proc_main(self):
        self._in_worker = True   # results leave through _deliver: proc_before_deliver orders them on the current stream
        super().proc_main()

    def _shared_flat(self, name: str, desc, loader):
        """The flat fp32 state_dict blob of one model on every worker of the node: rank 0 runs ``loader`` (reads the checkpoint(s),
        applies the key remaps / the DNI blend, or generates), the others receive it over the group.  Kept on the host for the second
        job set.  A loader that fails on rank 0 fails every rank (they would otherwise wait in the broadcast for ever)."""
        from .. import _capi
        n = _capi.param_count(desc)
        flat, err = None, None
        if self.node_rank == 0:
            try:
                flat = loader()
                if flat.size != n:
                    raise ValueError(f"{name}: the weights hold {flat.size} scalars, the model description needs {n}")
            except Exception as e:  # noqa: BLE001 - re-raised below, on every rank
                err = e
        if self.node_world > 1:
            import torch.distributed as dist
            msg = [None if err is None else f"{type(err).__name__}: {err}"]
            dist.broadcast_object_list(msg, src=0)
            if msg[0] is not None and err is None:
                raise RuntimeError(f"rank 0 could not load the {name} weights: {msg[0]}")
        if err is not None:
            raise err
        flat = sharding.broadcast_weights(flat, n, self.torch_device, force_collective=bool(self.group and self.group.force))
        self._flats[name] = (desc, flat)
        return flat

    def proc_init(self):
        log("proc init")
        self._open_device()
        with self._node_group():
            self._build_models()   # (weights resolved on rank 0 only, see _shared_flat)
        self._init_job_sets()
        self._init_host_io()
        log("model loaded")

    def _open_device(self):
        from .. import _capi
        self.ctx = _capi.Context(self.device)
        self.torch_device = self.ctx.device
        self._local_device = self.ctx.device_index

    @contextlib.contextmanager
    def _node_group(self):
        """Inside: this worker is rank ``node_rank`` of ``node_world`` (the group of ``self.group``, or one the process already sits
        in - ``bench.py`` under ``torch.distributed.run``).  A group joined here is left on the way out, error or not, unless
        ``group.keep``: once the weights are in place the workers of a node are independent processes."""
        import torch.distributed as dist
        had_group = dist.is_available() and dist.is_initialized()
        self._flats = {}
        self.node_rank, self.node_world = sharding.join_group(self.group, getattr(self, "_local_device", None))
        if self.node_world > 1:
            log(f"worker {self.node_rank} of {self.node_world} on {self.torch_device} ({dist.get_backend()})")
        try:
            yield
        finally:
            if not had_group and not (self.group and self.group.keep):
                sharding.leave_group()

    def _build_models(self):
        from .. import _capi
        from . import model as factory
        if self.weights == "synthetic":
            log("WARNING: weights='synthetic' - every network runs on generated weights, output frames are noise")
        def spec(name):
            return "synthetic" if self.weights == "synthetic" else (self.weights or {}).get(name)
        if self.upscaler_model == "fsrcnn":
            desc = factory.fsrcnn_desc(self.scale, self.fsrcnn_dtype, self.model_flags)
            flat = self._shared_flat("sr", desc, lambda: factory.fsrcnn_flat(self.scale, spec("sr"), self.seed, self.checkpoint_dir))
        else:
            name = self.model_name or factory.DEFAULT_REALESRGAN
            desc = factory.esrgan_desc(name, self.dtype, self.model_flags)
            flat = self._shared_flat("sr", desc, lambda: factory.esrgan_flat(
                name, self.denoise_rate, spec("sr"), self.seed, spec("sr_wdn") if self.weights != "synthetic" else None, self.checkpoint_dir))
        self.model = _capi.Model(self.ctx, desc, flat)
        self.denoise_model = None
        # quirk kept from the reference: with 'realesrgan' the batched path never denoises even when
        # denoising=True (fsrcnn_upscaler.py:109,168-233); the BSVD model is only used per-frame.
        if self.denoising and self.single_mode:
            ddesc = factory.denoise_desc(self.dtype)
            dflat = self._shared_flat("denoise", ddesc, lambda: factory.denoise_flat(spec("denoise"), self.seed, checkpoint_dir=self.checkpoint_dir))
            self.denoise_model = _capi.Model(self.ctx, ddesc, dflat)

    def _init_job_sets(self):
        # job sets: [0] is what every job ran on before; [1] (second context / model / stream) is built on the first one-frame job
        self._sets = [{"ctx": self.ctx, "model": self.model, "denoise": self.denoise_model, "up": None, "key": None, "stream": None}]
        self._alt = 0
        self._pending = {}
        self._inflight = collections.deque()   # (end-of-job event, input frames) of jobs on a job set's stream: see _retire
        self._streams_checked = False
        self._small = {}
        if not (self.overlap_jobs and not self.single_mode):
            self._flats.pop("sr", None)      # (no second set will ever be built: drop the host copy)
        self._flats.pop("denoise", None)     # (the batched path never denoises)
        # the most results the worker holds back (BaseService: held results).  While one-frame jobs alternate over the job sets, result i may
        # wait until jobs i + 1 .. i + sets - 1 have been enqueued - the wait for result i goes onto the current stream, which every later job's
        # stream waits for before it starts, and a consumer callback may block on it - but it leaves as soon as its end event has fired
        # (proc_result_ready).  A job that does not alternate (multi-frame: frame lanes on set 0) sets the lag to 0 and flushes (proc_deliver_lag).
        self.deliver_lag = self.overlap_sets - 1 if self._overlap_active() else 0
        self._lag_now = 0

    def _init_host_io(self):
        self._peer_copies = 0
        self._host_jobs = 0
        if not hasattr(self, "_pending"):
            self._pending = {}
        if self.host_rings is None:
            return
        self._host_on_gpu = self.torch_device.type == "cuda"   # (a CPU double of the worker - tests - reads and writes the rings directly)
        if not self._host_on_gpu:
            return
        for ring in self.host_rings:
            try:
                ring.pin()
            except RuntimeError as e:   # (copies to and from pageable memory still work - staged by the runtime, no longer asynchronous)
                log(f"WARNING: {e}: host frames go through UNPINNED memory")
        self._s_in, self._s_out = torch.cuda.Stream(self.torch_device), torch.cuda.Stream(self.torch_device)
        self._stage = {}    # input shape -> [[device tensor, event after which it may be overwritten], ...], used round robin
        self._stage_at = {}
        log(f"host frame rings pinned: {self.host_rings[0].slots} slots, {self.host_rings[0].slot_bytes >> 10} KB in / {self.host_rings[1].slot_bytes >> 10} KB out each")

    def _staging(self, shape):
        """A device tensor for one job's input frames.  More of them than jobs can be in flight, each guarded by the end event of the job
        that last read it (the copy stream waits for it before it overwrites the tensor)."""
        bufs = self._stage.setdefault(shape, [])
        depth = self.overlap_sets + 2
        if len(bufs) < depth:
            bufs.append([torch.empty(shape, dtype=torch.uint8, device=self.torch_device), None])
            return bufs[-1]
        i = self._stage_at.get(shape, 0)
        self._stage_at[shape] = (i + 1) % depth
        return bufs[i]

    def _host_job(self, hf: HostFrames):
        """H2D on the copy stream -> upscale -> D2H on the other copy stream; returns (result shape, event after which the result slot holds
        the frames - None when it already does)."""
        src = self.host_rings[0].view(hf.slot, hf.shape)
        if not self._host_on_gpu:
            out = self.upscale(src)
            self.host_rings[1].view(hf.out_slot, tuple(out.shape)).copy_(out)
            return tuple(out.shape), None
        cur = torch.cuda.current_stream(self.torch_device)
        stage = self._staging(tuple(hf.shape))
        with torch.cuda.stream(self._s_in):
            if stage[1] is not None:
                self._s_in.wait_event(stage[1])
            stage[0].copy_(src, non_blocking=True)
            copied = self._s_in.record_event()
        cur.wait_event(copied)
        out = self.upscale(stage[0])
        rec = self._pending.pop(id(out), None)           # (job-set path: the result is ordered on its set's stream, not on the current one)
        done = rec[1] if rec is not None else cur.record_event()
        stage[1] = done
        with torch.cuda.stream(self._s_out):
            self._s_out.wait_event(done)
            self.host_rings[1].view(hf.out_slot, tuple(out.shape)).copy_(out, non_blocking=True)
            landed = self._s_out.record_event()
        out.record_stream(self._s_out)
        return tuple(out.shape), landed

    def proc_job_recieved(self, job):
        if not isinstance(getattr(job, "frames", None), HostFrames):
            entry = super().proc_job_recieved(job)
            return self._to_host_result(entry) if self.host_results and getattr(self, "_in_worker", False) else entry
        # the same spans as BaseUpscalerService.proc_job_recieved, around the host job
        import time
        hf = job.frames
        if self.host_rings is None:
            raise RuntimeError("a HostFrames job reached a worker that was started without host rings")
        prof = job.profiler if getattr(job, "profiler", None) is not None else Profiler()
        self.profiler = prof
        arrived = time.time()
        prof.end("recoder.output")
        prof.start("upscaler.upscale")
        try:
            shape, landed = self._host_job(hf)
        finally:
            prof.end("upscaler.upscale")
        self._host_jobs += 1
        answer_frames = HostFrames(slot=hf.slot, out_slot=hf.out_slot, shape=shape, result=True)
        if landed is not None:
            self._lag_now = max(getattr(self, "_lag_now", 0), 1)   # one more job is enqueued before this result is waited for: the copies hide under it
            self._pending[id(answer_frames)] = (answer_frames, landed, "host")
        prof.set("upscaler.input.peer_copies", self._peer_copies)
        prof.start("upscaler.output")
        return answer(job, answer_frames, time.time() - arrived, prof)

    def start(self) -> None:
        if self.host_results and self.result_ring is None:   # (here, in the caller's process: the ring must exist before the worker does)
            from ..hostring import HostRing
            oh, ow = self.out_hw(*self.lr_shape)
            nbytes = self.host_result_bytes or max(1, int(self.batch_size)) * oh * ow * 3
            self.result_ring = HostRing(self.HOST_RESULT_SLOTS, nbytes, "ss4k_results")
        super().start()

    def _to_host_result(self, entry):
        """``host_results``: replace the entry's device frames by a view of the next slot of the pinned result ring (D2H on the worker's
        copy stream, ordered after the job's end event; the entry is held until the bytes have landed - ``proc_result_ready`` /
        ``proc_before_deliver``)."""
        from ..hostring import RingView
        out = entry.frames
        ring = self.result_ring
        if ring is None or not (isinstance(out, torch.Tensor) and out.is_cuda):
            return entry
        if not ring.fits(out.shape):
            if not getattr(self, "_warned_big", False):
                self._warned_big = True
                log(f"host_results: a result of {tuple(out.shape)} does not fit a {ring.slot_bytes}-byte slot (host_result_bytes) - it stays on the device")
            return entry
        if not hasattr(self, "_s_out"):
            self._s_out = torch.cuda.Stream(self.torch_device)
        if not getattr(self, "_result_ring_pinned", False):
            self._result_ring_pinned = True
            try:
                ring.pin()
            except RuntimeError as e:
                log(f"WARNING: {e}: host results go through UNPINNED memory")
        slot = self._result_slot = (getattr(self, "_result_slot", -1) + 1) % ring.slots
        dst = ring.view(slot, tuple(out.shape))
        rec = self._pending.pop(id(out), None)
        cur = torch.cuda.current_stream(self.torch_device)
        done = rec[1] if rec is not None else cur.record_event()
        with torch.cuda.stream(self._s_out):
            self._s_out.wait_event(done)
            dst.copy_(out, non_blocking=True)
            landed = self._s_out.record_event()
        out.record_stream(self._s_out)
        # on_queue runs in this process: it gets the tensor; the result queue gets the 100-byte handle that unpickles as the same view
        entry.frames = dst if self.on_queue is not None else RingView(ring, slot, out.shape)
        self._pending[id(entry.frames)] = (entry.frames, landed, "host")
        self._lag_now = max(getattr(self, "_lag_now", 0), 1)
        return entry

    def _overlap_active(self) -> bool:
        return bool(self.overlap_jobs) and not self.single_mode and "sr" in getattr(self, "_flats", {})

    def _job_set(self, k: int) -> dict:
        from .. import _capi
        while len(self._sets) <= k:
            desc, flat = self._flats["sr"]
            ctx = _capi.Context(self.device)
            self._sets.append({"ctx": ctx, "model": _capi.Model(ctx, desc, flat), "denoise": None, "up": None, "key": None, "stream": None})
            log(f"job set {len(self._sets) - 1} built (one-frame jobs alternate over {len(self._sets)} sets)")
        js = self._sets[k]
        if js["stream"] is None and self._overlap_active():
            others = [torch.cuda.current_stream(self.torch_device)] + [o["stream"] for o in self._sets if o["stream"] is not None]
            js["stream"] = self._vetted_stream(others, f"job set {k}")
        return js

    def _vetted_stream(self, others, what: str, tries: int = 8) -> torch.cuda.Stream:
        """A new stream that passes the library's pair test (``ss4k_stream_pair_check``, ~ 3 ms a pair) against every stream of `others`: HIP
        serves a process's streams from a few hardware queues, two streams of one queue run in order, and some PAIRS of queues are slow while
        both are busy - or while one is merely WAITING for the other, as the current stream does for a job set's (measured: a one-frame job
        17.5 instead of 9.5 ms on a set-0 stream that was the process's 4th, `profiles/earlier/r05/r05_lane_queue.txt`).  Which queue a stream gets depends
        on how many the process created before."""
        if under_profiler():
            return torch.cuda.Stream(self.torch_device)
        for _ in range(tries):
            cand = torch.cuda.Stream(self.torch_device)   # (torch hands out the streams of a fixed pool in turn: always a different one)
            bad = next((o for o in others if not self.ctx.streams_side_by_side(o, cand)), None)
            if bad is None:
                return cand
            log(f"{what}: stream {cand.cuda_stream:#x} fails the pair test against {bad.cuda_stream:#x} - taking another")
        log(f"{what}: no stream passed the pair test after {tries} tries; using the last one")
        return cand

    #: a job alternates over the job sets only if the network's activation workspace for it is at most this (RRDBNet x2 on a 720p frame: 1.25 GB -
    #: 0.19 GB of body tensors at 360 x 640 and 1.06 GB of tail tensors at 720p / 1440p -, on a 1080p frame 2.8 GB; RRDBNet x4 on 1080p: 11 GB:
    #: 4050 tiles per launch fill the chip eight times over, the next job has nothing to cover (measured: 14.1 frames/s either way) and two more
    #: sets would be 22 GB for nothing)
    SMALL_JOB_WORKSPACE = 4 << 30

    def _small_job(self, frames: torch.Tensor) -> bool:
        key = tuple(frames.shape[:3])
        if key not in self._small:
            if len(self._small) > 256:   # an image server fed arbitrary sizes must not grow this without bound
                self._small.clear()
            n, h, w = key
            if self.lr_hr_resize and (h > self.lr_shape[0] or w > self.lr_shape[1]):
                h, w = self.lr_shape   # (the network sees the area-resized frame: fsrcnn_upscaler.py:174-176)
            try:
                self._small[key] = self.model.workspace_bytes(n, h, w) <= self.SMALL_JOB_WORKSPACE
            except Exception:  # noqa: BLE001 - a shape the network rejects: the job itself reports it
                self._small[key] = False
        return self._small[key]

    def _check_streams(self, frames: torch.Tensor, tries: int = 4) -> None:
        """One-off, at the first small job: do the job sets' streams really run side by side?  HIP serves a process's streams from a few
        hardware queues; two streams that share one are executed in order, whatever the program says - seen on a process that had created
        many streams before (profiles/earlier/r05/r05_n1_probe_streams.txt: 107 instead of 122 frames/s, silently).  So, for every pair of sets: six
        one-frame jobs alternating over the two, first with both sets on ONE stream, then each on its own; side by side they take 0.88-0.92 of
        the time in order (what is gained is the overlap of a job's tail with the next one's head, so a single pair of jobs shows only half of
        it), on a shared queue 0.97-1.2.  A set that does not pass takes another stream and is checked again.  Blocks the host for a few dozen
        jobs' time, once per service."""
        self._streams_checked = True
        if under_profiler():
            return
        frames = frames[:1]   # (one-frame jobs gain 12-15 % from running side by side: a clear signal; multi-frame jobs gain nothing, measured)
        dev, cur = self.torch_device, torch.cuda.current_stream(self.torch_device)
        sets = [self._job_set(k) for k in range(self.overlap_sets)]
        ups = [self._get_upscaler(k) for k in range(self.overlap_sets)]

        def run(seq, stream_of):   # jobs on the sets of `seq`, in that order, set i on stream_of[i]: milliseconds on the current stream
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            used = {stream_of[i] for i in seq}
            for st in used:
                st.wait_stream(cur)
            for i in seq:
                with torch.cuda.stream(stream_of[i]):
                    ups[i](frames)
            for st in used:
                cur.wait_stream(st)
            e1.record(cur)
            e1.synchronize()
            return e0.elapsed_time(e1)

        run(range(len(sets)), {i: js["stream"] for i, js in enumerate(sets)})   # (first calls size workspaces and raise LDS limits)
        for k in range(1, len(sets)):
            for _ in range(tries):
                bad = None
                for j in range(k):   # against EVERY earlier set: sets 1 and 2 must not share a queue either
                    seq = (j, k) * 3
                    serial = run(seq, {j: sets[j]["stream"], k: sets[j]["stream"]})
                    both = run(seq, {j: sets[j]["stream"], k: sets[k]["stream"]})
                    if not both < 0.95 * serial:
                        bad = (j, both, serial)
                        break
                if bad is None:
                    log(f"job set {k}: runs beside the earlier sets (six jobs {both:.2f} ms against {serial:.2f} ms in order)")
                    break
                log(f"job set {k}: its stream does not run beside set {bad[0]}'s (six jobs {bad[1]:.2f} ms against {bad[2]:.2f} ms in order) - taking another stream")
                sets[k]["stream"] = self._vetted_stream([cur] + [sets[j]["stream"] for j in range(k)], f"job set {k}")
            else:
                log(f"job set {k}: no stream found that overlaps with the earlier sets' after {tries} tries; its jobs will run in order with one of them")

    def _get_upscaler(self, k: int = 0):
        from .. import _capi
        js = self._job_set(k)
        key = (tuple(self.lr_shape), None if self.output_shape is None else tuple(self.output_shape),
               bool(self.lr_hr_resize), bool(self.single_mode), float(self.denoise_rate))
        if js["up"] is None or key != js["key"]:
            js["up"] = _capi.Upscaler(js["ctx"], js["model"], self.lr_shape, self.output_shape, self.lr_hr_resize,
                                      self.single_mode, js["denoise"], self.denoise_rate)
            js["key"] = key
        return js["up"]

    def _retire(self) -> None:
        """Let go of the input frames of jobs whose device work has finished.  A job's input is usually a tensor received over CUDA IPC; when
        the worker's last reference to it dies, the PRODUCER process gets the block back at once and may hand it to its next frame - nothing
        orders that against this worker's streams - while the job (enqueued, not finished) still reads it.  So the service keeps the reference
        until the job's end event has fired (tools/service_soak.py --no-hold shows the race: 3 wrong results in 1500 jobs without this)."""
        q = getattr(self, "_inflight", None)
        while q and q[0][0].query():
            q.popleft()

    def proc_before_deliver(self, entry):
        self._retire()
        # a result computed on a job set's own stream: the current stream - on which the result tensor is handed to the consumer
        # (on_queue, or the IPC event torch records when the tensor is pickled into the result queue) - waits for it here
        pending = getattr(self, "_pending", None)
        rec = pending.pop(id(entry.frames), None) if pending and getattr(entry, "frames", None) is not None else None
        if rec is not None and len(rec) > 2:
            rec[1].synchronize()     # a host result: the consumer reads host memory (a ring slot, possibly from another process) - the bytes must have landed
        elif rec is not None:
            torch.cuda.current_stream(self.torch_device).wait_event(rec[1])

    def proc_deliver_lag(self) -> int:
        return getattr(self, "_lag_now", 0)   # set by upscale() from the path the NEWEST job took

    def proc_result_ready(self, entry) -> bool:
        pending = getattr(self, "_pending", None)
        rec = pending.get(id(entry.frames)) if pending and getattr(entry, "frames", None) is not None else None
        return rec is None or rec[1].query()

    def proc_cleanup(self):
        pass

    def _run(self, k: int, frames: torch.Tensor) -> torch.Tensor:
        up = self._get_upscaler(k)
        out = up(frames)
        prof = getattr(self, "profiler", None)
        if prof is not None:
            # the reference's span keys (fsrcnn_upscaler.py:276-278,290-300): host time around the
            # asynchronous stage launches, measured inside the library
            denoise_ms, model_ms = up.last_enqueue_ms()
            if self._sets[k]["denoise"] is not None:
                record_span(prof, "fsrcnn.denoise", denoise_ms / 1000.0)
            record_span(prof, "fsrcnn.model", model_ms / 1000.0)
        return out

    def upscale(self, frames: torch.Tensor, wait: bool = True):
        """``wait=False`` (direct callers only): with the one-frame overlap active the result is ordered on its job set's stream, not
        on the current one - synchronise (``torch.cuda.synchronize()``) before reading it.  The default orders it on the current
        stream like any torch op."""
        assert isinstance(frames, torch.Tensor)
        if frames.device != self.torch_device:
            if frames.is_cuda:   # a device tensor from ANOTHER GPU (the caller put job s somewhere else than on GPU s % G): one peer copy, counted
                self._peer_copies = getattr(self, "_peer_copies", 0) + 1
                prof = getattr(self, "profiler", None)
                if prof is not None:
                    prof.set("upscaler.input.peer_copies", self._peer_copies)
            frames = frames.to(self.torch_device, non_blocking=True)
        if frames.ndim != 4:
            raise Exception(frames.shape)
        assert frames.shape[-1] == 3
        self._lag_now = 0
        if not self._overlap_active():
            out = self._run(0, frames)
            if frames.is_cuda and hasattr(self, "_inflight"):   # (same hold on the input as below: see _retire)
                self._retire()
                self._inflight.append((torch.cuda.current_stream(self.torch_device).record_event(), frames))
            return out
        k = 0
        if frames.shape[0] <= self.overlap_max_frames and self._small_job(frames):   # consecutive one-frame jobs alternate; a multi-frame job overlaps with itself (frame lanes) on set 0
            if not self._streams_checked:
                self._check_streams(frames)
            k, self._alt = self._alt, (self._alt + 1) % self.overlap_sets
            self._lag_now = self.overlap_sets - 1   # (only while jobs alternate: the next ones have something to overlap with)
        cur = torch.cuda.current_stream(self.torch_device)
        side = self._job_set(k)["stream"]
        side.wait_stream(cur)   # the frames (an IPC tensor's ready event, a .to(device) copy) are ordered on the current stream
        frames.record_stream(side)
        with torch.cuda.stream(side):
            out = self._run(k, frames)
        done = side.record_event()
        self._retire()
        self._inflight.append((done, frames))
        out.record_stream(cur)
        if getattr(self, "_in_worker", False):
            self._pending[id(out)] = (out, done)   # proc_before_deliver makes the current stream wait, one job later
        elif wait:
            cur.wait_event(done)
        return out
