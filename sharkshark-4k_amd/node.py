"""``UpscalerNode``: the multi-GPU form of the service - one call, G workers, one per GPU.

The reference builds ONE upscaler service on ``device=0`` and feeds it from the stream pipeline
(``src/sharkshark/pipeline.py:15-50,61-149``).  Frames (jobs) are independent, so a node of G MI355X runs G of
them, and this class is everything an integrator needs around that::

    node = UpscalerNode(devices=8, upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False,
                        checkpoint_dir="/models")          # same keyword arguments as HipUpscalerService
    node.start()                                           # G spawned workers; rank 0 loads, RCCL broadcast, all ready
    steps = node.submit_batch(frames_u8_nhwc)              # 1 s of HOST frames (numpy / CPU tensor) -> jobs of min(4, fps) frames, job `step` -> GPU step % G
    for entry in node.poll(timeout=0.1): sink(entry)       # results re-ordered by step; entry.frames = a view of pinned host memory, valid until the next poll()
    node.stop()

What it does, in order:

* builds G service objects (``service_cls(device=k, group=GroupSpec(rank=k, world=G, port), **kw)``) and starts each as a FRESH
  SPAWNED child (``BaseService.start``; the parent never needs a HIP context, and no process that touched a GPU is ever re-exec'd);
* worker k's ``proc_init`` joins the node's process group (``nccl`` = RCCL over xGMI when every worker owns its own GPU, ``gloo``
  when two workers share one or on CPU), ONLY RANK 0 reads / repacks / blends the checkpoints, ``sharding.broadcast_weights`` hands
  every blob to the others, and the group is left again: no collective ever runs on the data path;
* ``start()`` returns when every worker has reported ready (``BaseService.ready_event``) - or raises if one died during start-up;
* HOST FRAMES are the node's native input (``hostring.py``; the reference's recorder / streamer processes hold host numpy frames,
  ``pipeline.py:84-93``, ``streamer.py:92-98``): per worker, two rings of ``host_slots`` pinned shared-memory slots sized for one job of
  ``host_frames`` = (H, W) frames in and its result out are created HERE, once, before the worker starts; worker k copies job frames to
  and from GPU k on its own copy streams with two jobs in flight; the parent never opens a HIP context.  A device tensor given to
  ``submit_batch`` passes through as before (one on the wrong GPU is copied over by the worker and counted in ``report()['peer_copies']``);
* a ``StreamDispatcher`` fans jobs out ``step % G`` over the LIVING workers and re-orders results by ``step``;
* a worker that dies later is routed around at once (the node runs on G - 1; the jobs that were inside it are counted in
  ``report()['lost']``); ``replace_dead()`` starts a new child in its slot - a world-of-one worker that loads the weights itself, as
  the group of the start-up is gone by then.  ``auto_replace=True`` does that by itself, without blocking the stream: ``poll()`` starts
  the replacement when it sees a dead worker and swaps it in once it reports ready (the reference's image server restarts its pipeline on a
  dead worker, ``image_pipeline.py:66-73,295-301``; its deployment script loops ``while true``).
"""
from __future__ import annotations

import socket
import time
from typing import List, Optional, Sequence, Union

from . import hostring, sharding
from .stream import StreamDispatcher
from .upscale.hip_upscaler import HipUpscalerService


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class UpscalerNode:
    def __init__(self, devices: Union[int, Sequence[int], None] = None, service_cls=HipUpscalerService, backend: Optional[str] = None,
                 fps: int = 24, frame_skips: bool = True, on_result=None, lost_after_s: float = 5.0, force_group: bool = False,
                 output_shape="unset", host_frames=True, host_slots: int = 6, auto_replace: bool = False, **service_kwargs):
        if devices is None:
            import torch
            devices = torch.cuda.device_count()   # (counting devices does not initialise the GPU in this process)
        self.devices: List[int] = list(range(devices)) if isinstance(devices, int) else [int(d) for d in devices]
        if not self.devices:
            raise ValueError("UpscalerNode needs at least one device")
        if backend is None and len(set(self.devices)) < len(self.devices):
            backend = "gloo"   # RCCL refuses two ranks on one GPU: workers that share a device exchange the weights through host memory
        self.service_cls, self.service_kwargs, self.backend, self.force_group = service_cls, dict(service_kwargs), backend, force_group
        self.output_shape = output_shape   # the pipelines overwrite this attribute on the service object (pipeline.py:46-50)
        # host-frame rings per worker: True = sized for jobs of min(4, fps) frames of the service's lr_shape; (H, W) = of that frame size
        # (a recorder that delivers frames bigger than lr_shape, which the service area-resizes); None / False = no rings
        self.host_frames, self.host_slots, self.job_frames = host_frames, int(host_slots), min(4, int(fps))
        self.port = _free_port()
        self.services = [self._make_service(k, len(self.devices)) for k in range(len(self.devices))]
        self.dispatcher = StreamDispatcher(self.services, fps=fps, frame_skips=frame_skips, on_result=on_result, lost_after_s=lost_after_s)
        self.started = False
        self.auto_replace = bool(auto_replace)
        self._replacing = {}      # slot -> replacement service that has been started and is not ready yet
        self.replaced_total = 0

    def _make_service(self, k: int, world: int):
        import torch.multiprocessing as mp
        group = sharding.GroupSpec(rank=k if world > 1 else 0, world=world, master_port=self.port, backend=self.backend,
                                   force=self.force_group and world == 1)
        svc = self.service_cls(device=self.devices[k], group=group, **self.service_kwargs)
        if self.output_shape != "unset":
            svc.output_shape = self.output_shape
        if self.host_frames:
            h, w = svc.lr_shape if self.host_frames is True else self.host_frames
            oh, ow = svc.out_hw(h, w)
            svc.host_rings = hostring.make_rings(self.host_slots, self.job_frames * h * w * 3, self.job_frames * oh * ow * 3)
        svc.mp_start_method = "spawn"   # the launcher may have touched the GPU (warm-up, device queries); workers are fresh interpreters
        svc.ready_event = mp.get_context("spawn").Event()
        return svc

    # ------------------------------------------------------------------------------------------
    def start(self, timeout: float = 600.0) -> "UpscalerNode":
        for svc in self.services:
            svc.start()
        self.started = True
        self._wait_ready(self.services, timeout)
        return self

    @staticmethod
    def _wait_ready(services, timeout: float) -> None:
        deadline = time.monotonic() + timeout
        waiting = list(services)
        while waiting:
            for svc in list(waiting):
                if svc.ready_event.wait(0.05):
                    waiting.remove(svc)
                elif not svc.proc.is_alive():
                    raise RuntimeError(f"UpscalerNode: the worker on device {svc.device} died during start-up (exit code {svc.proc.exitcode})")
            if waiting and time.monotonic() > deadline:
                raise TimeoutError(f"UpscalerNode: {len(waiting)} worker(s) not ready after {timeout:.0f} s")

    def alive(self) -> List[bool]:
        return [svc.proc.is_alive() for svc in self.services]

    def replace_dead(self, timeout: float = 600.0) -> List[int]:
        """Start a fresh child in the slot of every dead worker; returns the slots replaced.  The replacement is a world-of-one worker
        (it resolves ``weights`` itself): the start-up group no longer exists."""
        slots = [k for k, ok in enumerate(self.alive()) if not ok]
        fresh = []
        for k in slots:
            svc = self._make_service(k, 1)
            svc.start()
            fresh.append(svc)
        self._wait_ready(fresh, timeout)
        self.dispatcher.rescue_orphans()   # (host steps still inside the dead workers' rings: re-queued before those rings go)
        for k, svc in zip(slots, fresh):
            old = self.services[k]
            self.services[k] = svc
            self.dispatcher.services[k] = svc
            self._close_rings(old)
        return slots

    @staticmethod
    def _close_rings(svc) -> None:
        for ring in getattr(svc, "host_rings", None) or ():
            ring.close()

    # ------------------------------------------------------------------------------------------ the stream caller's interface
    def submit_batch(self, frames, audio_segment=None, profiler=None):
        return self.dispatcher.submit_batch(frames, audio_segment, profiler)

    def poll(self, timeout: float = 0.0):
        if self.auto_replace and self.started:
            self._auto_replace_step()
        return self.dispatcher.poll(timeout)

    def _auto_replace_step(self) -> None:
        """Non-blocking: start a replacement for every dead worker that has none yet; swap in the ones that have reported ready; start
        another one for a replacement that died while loading."""
        for k, svc in enumerate(self.services):
            if k in self._replacing:
                new = self._replacing[k]
                if new.ready_event.is_set():
                    self.dispatcher.rescue_orphans()     # (host steps still in the dead worker's ring, before it goes)
                    old = self.services[k]
                    self.services[k] = new
                    self.dispatcher.services[k] = new
                    self._close_rings(old)
                    del self._replacing[k]
                    self.replaced_total += 1
                elif not new.proc.is_alive():
                    self._close_rings(new)
                    del self._replacing[k]                # (it died during start-up: the next poll starts another)
            elif not svc.proc.is_alive():
                new = self._make_service(k, 1)
                new.start()
                self._replacing[k] = new

    def drain(self, expected_steps, timeout: float = 60.0):
        return self.dispatcher.drain(expected_steps, timeout)

    def report(self) -> dict:
        r = self.dispatcher.report()
        r["alive"] = self.alive()
        r["replaced"] = self.replaced_total
        r["replacing"] = sorted(self._replacing)
        return r

    def stop(self) -> List[Optional[int]]:
        codes = []
        for new in list(self._replacing.values()):   # (replacements that were still loading)
            if new.proc.is_alive():
                new.proc.kill()
                new.proc.join(timeout=15)
            self._close_rings(new)
        self._replacing.clear()
        for svc in self.services:
            if svc.proc.is_alive():
                try:
                    svc.stop()
                except Exception:  # noqa: BLE001 - a worker that dies between the check and the command is already stopped
                    pass
                if svc.proc.is_alive():
                    # it did not take the exit command (e.g. still inside a collective whose peer died during start-up): end exactly
                    # this child - the process object this node started
                    svc.proc.kill()
                    svc.proc.join(timeout=15)
            codes.append(svc.proc.exitcode)
        return codes

    def close(self) -> None:
        """Give the host rings back (after ``stop()``; views handed out by ``poll()`` die with them)."""
        for svc in self.services:
            self._close_rings(svc)

    def __enter__(self):
        return self.start() if not self.started else self

    def __exit__(self, *exc):
        self.stop()
        return False
