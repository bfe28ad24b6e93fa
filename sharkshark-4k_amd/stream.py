"""Stream-mode caller of the upscaler service(s): the hot path's immediate neighbour.

Restates what the reference's ``TwitchUpscalerPostStreamer`` does around the upscaler
(``src/sharkshark/pipeline.py:61-149``), generalised from one GPU to G:

* a recorder batch (1 s of frames) is cut into jobs of ``small_batch_size = min(4, fps)`` frames
  (``pipeline.py:31,80-84``), each tagged with a monotonically increasing ``step``;
* job ``step`` goes to service ``step % G`` with ``push_job_nowait``; a full queue drops the job
  (frame-skip back-pressure, ``pipeline.py:103-108``) unless ``frame_skips=False``;
* results come back per service in any interleaving and are re-ordered by ``step`` before they
  are handed to the sink (the reference only warns on out-of-order steps, ``streamer.py:77-78``);
* the sink gets ``'upscaler.upscale.per_frame_ms'`` and queue depths in the profiler
  (``pipeline.py:140-149``);
* a worker process that has died is routed around at once: its steps go to the next living service (the node runs on G - 1), and a
  step whose result can no longer come back because the worker it was queued in is gone is declared lost without waiting;
* HOST frames (numpy arrays / CPU tensors) go through the target worker's pinned rings when it has them (``hostring.py``, allocated by
  ``node.UpscalerNode``): the job is copied into a free input slot, a ~ 100-byte descriptor travels through the queue, the worker moves
  the frames to and from ITS GPU on its own copy streams, and the result handed to the sink is a zero-copy view of the worker's output
  ring, VALID UNTIL THE NEXT ``poll()`` (``drain()`` hands out copies).  No free slot counts as a full queue (frame skip), or - with
  ``frame_skips=False`` - is waited for while finished results are copied out of the rings to make room.  Device tensors pass through
  untouched (a tensor on another GPU than the worker's is copied over once by the worker: ``report()['peer_copies']``); host frames for a
  worker without rings, or bigger than a slot, travel as pickled tensors the old way (``report()['host_fallback']``);
* a HOST job that was inside a worker when it died is not lost: its frames are still in that worker's input ring (shared memory the
  parent holds), so the step is copied into a living worker's ring and queued again (``report()['rescued']``; results still leave in step
  order).  Device-tensor jobs have no such copy and are counted lost, as before;
* a result that never arrives (a worker died, or ``BaseService`` dropped it on a full result queue)
  must not stall a 24/7 stream: a step that keeps later results waiting for more than
  ``lost_after_s`` seconds, or behind more than ``max_reorder`` pending results, is declared lost,
  counted in ``report()['lost']`` and skipped.
"""
from __future__ import annotations

import math
import queue
import sys
import time
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .hostring import HostFrames, SlotPool
from .upscale.upscaler_base import UpscalerQueueEntry
from .util.profiler import Profiler


class StreamDispatcher:
    def __init__(self, services: Sequence, fps: int = 24, frame_skips: bool = True,
                 on_result: Optional[Callable[[UpscalerQueueEntry], None]] = None, max_reorder: int = 64,
                 lost_after_s: float = 5.0):
        assert len(services) >= 1
        self.services = list(services)
        self.fps = fps
        self.small_batch_size = min(4, int(fps))
        self.frame_skips = frame_skips
        self.on_result = on_result
        self.frame_step = 0
        self.next_emit = 0
        self._dropped = set()        # steps skipped at submit time and not yet passed by next_emit
        self.dropped_total = 0
        self.lost_total = 0          # steps that were queued but whose result never came back
        self.late_total = 0          # results that arrived after their step had been declared lost (discarded)
        self._pending: Dict[int, UpscalerQueueEntry] = {}
        self._owner: Dict[int, object] = {}   # queued step -> the service it went to (until its result is back or it is given up)
        self.rerouted_total = 0      # steps that went to another service than step % G because that one's worker was dead
        self.max_reorder = max_reorder
        self.lost_after_s = lost_after_s
        self._stalled_since: Optional[float] = None
        self.last_reported = time.time()
        # host-frame rings: slot accounting per worker (keyed by the service object), the slots a queued step holds, and the output slots
        # whose views the last poll() handed out (given back at the start of the next one)
        self._slots: Dict[object, tuple] = {}     # step -> (service, input slot, output slot); ('view', step) -> (service, output slot)
        self._lent: List[tuple] = []              # (service, output slot) behind the views the last poll() handed out
        self._lent_steps = set()
        self.host_jobs_total = 0
        self.host_fallback_total = 0
        self.rescued_total = 0
        self._jobs: Dict[int, tuple] = {}         # queued host step -> (shape, audio, profiler, rescues so far): what a re-submission needs
        self._peer_copies: Dict[int, int] = {}    # service index -> the worker's running count, from its latest result
        self.push_timeout = 10.0

    @property
    def dropped(self) -> List[int]:
        """Dropped steps the ordered emission has not passed yet (older ones are pruned)."""
        return sorted(self._dropped)

    @staticmethod
    def _alive(svc) -> bool:
        """A service whose worker process was started and has exited is dead; doubles without a process are always alive."""
        proc = getattr(svc, "proc", None)
        if proc is None or proc.is_alive():
            return True
        return proc.exitcode is None and getattr(proc, "pid", None) is None   # (not started yet)

    def _pick(self, step: int):
        n = len(self.services)
        for j in range(n):
            svc = self.services[(step + j) % n]
            if self._alive(svc):
                if j:
                    self.rerouted_total += 1
                return svc
        raise RuntimeError("StreamDispatcher: no living upscaler worker")

    # ------------------------------------------------------------------------------------------ host-frame rings
    def _pool(self, svc) -> Optional[SlotPool]:
        rings = getattr(svc, "host_rings", None)
        if rings is None:
            return None
        pool = svc.__dict__.get("_slot_pool")      # (parent-side state kept on the service object: a replaced worker brings its own)
        if pool is None:
            pool = svc.__dict__["_slot_pool"] = SlotPool(rings[0].slots)
        return pool

    def _release_lent(self) -> None:
        for svc, out_slot in self._lent:
            self._pool(svc).give_out(out_slot)
        self._lent.clear()
        self._lent_steps.clear()

    def _own_result(self, e) -> None:
        """Turn a pending result that is still a ring view into a tensor of its own and give the slot back (slow path: only when the
        rings run out because the consumer is behind)."""
        held = self._slots.pop(("view", e.step), None)
        if held is not None:
            e.frames = e.frames.clone()
            self._pool(held[0]).give_out(held[1])

    def _take_slots(self, svc, wait: bool):
        pool = self._pool(svc)
        got = pool.take()
        if got is not None or not wait:
            return got
        deadline = time.monotonic() + self.push_timeout
        while got is None and time.monotonic() < deadline:   # (views the last poll() handed out stay valid: only pending results are moved)
            self._collect()
            for e in self._pending.values():          # finished results waiting for an earlier step: copy them out of the ring
                self._own_result(e)
            got = pool.take()
            if got is None:
                time.sleep(0.0005)
        return got

    # pipeline.py:61-108
    def submit_batch(self, frames, audio_segment=None, profiler: Optional[Profiler] = None) -> List[int]:
        """Cut one recorder batch into jobs and fan them out; returns the steps actually queued."""
        assert self.small_batch_size != 0
        profiler = profiler or Profiler()
        njobs = math.ceil(len(frames) / self.small_batch_size)
        queued = []
        for i in range(njobs):
            profiler.start("recoder.output.entry")
            chunk = frames[i * self.small_batch_size:(i + 1) * self.small_batch_size]
            audio = None
            if audio_segment is not None:
                per = len(audio_segment) // njobs
                audio = torch.as_tensor(audio_segment[i * per:(i + 1) * per])
            step = self.frame_step
            self.frame_step += 1
            svc = self._pick(step)
            slots = None
            try:
                on_host = not (isinstance(chunk, torch.Tensor) and chunk.is_cuda)
                if on_host and self._pool(svc) is not None and svc.host_rings[0].fits(chunk.shape):
                    slots = self._take_slots(svc, wait=not self.frame_skips)
                    if slots is None:
                        raise queue.Full
                    shape = svc.host_rings[0].write(slots[0], chunk)
                    payload = HostFrames(slot=slots[0], out_slot=slots[1], shape=shape)
                    self.host_jobs_total += 1
                else:
                    payload = torch.as_tensor(chunk)
                    shape = tuple(payload.shape)
                    self.host_fallback_total += int(on_host)
                entry = UpscalerQueueEntry(frames=payload, audio_segment=audio, step=step, profiler=profiler)
                profiler.set("recoder.output.frames.shape", str(shape))
                profiler.end("recoder.output.entry")
                if self.frame_skips:
                    svc.push_job_nowait(entry)
                else:
                    svc.push_job(entry, timeout=self.push_timeout)
                queued.append(step)
                self._owner[step] = svc
                if slots is not None:
                    self._slots[step] = (svc, slots[0], slots[1])
                    self._jobs[step] = (shape, audio, profiler, 0)
            except queue.Full:
                if slots is not None:
                    self._pool(svc).give_in(slots[0])
                    self._pool(svc).give_out(slots[1])
                self._dropped.add(step)
                self.dropped_total += 1
                print("StreamDispatcher: upscaler queue full, job skipped", file=sys.stderr)
        return queued

    def rescue_orphans(self) -> int:
        """Host steps whose worker has died (and whose results, if any, have been collected): their frames still sit in the dead worker's
        input ring - copy them into a living worker's ring and queue them again.  Returns how many were re-submitted."""
        n = 0
        for step, owner in list(self._owner.items()):
            held, job = self._slots.get(step), self._jobs.get(step)
            if held is None or job is None or self._alive(owner) or not owner.result_queue.empty() or owner.host_rings is None:
                continue
            shape, audio, profiler, tries = job
            if tries >= 3:
                continue
            try:
                svc = self._pick(step)
            except RuntimeError:
                return n          # nobody left alive
            pool = self._pool(svc)
            if pool is None or not svc.host_rings[0].fits(shape):
                continue
            slots = pool.take()
            if slots is None:
                continue          # (next poll)
            try:
                svc.host_rings[0].write(slots[0], owner.host_rings[0].view(held[1], shape))
                svc.push_job_nowait(UpscalerQueueEntry(frames=HostFrames(slot=slots[0], out_slot=slots[1], shape=shape), audio_segment=audio,
                                                       step=step, profiler=profiler))
            except (queue.Full, ValueError, TypeError):   # (a closed ring, a full queue: leave the step to the lost-step logic)
                pool.give_in(slots[0]); pool.give_out(slots[1])
                continue
            self._owner[step] = svc
            self._slots[step] = (svc, slots[0], slots[1])
            self._jobs[step] = (shape, audio, profiler, tries + 1)
            self.rescued_total += 1
            n += 1
        return n

    def _emit_ready(self, force: bool = False):
        out = []
        while True:
            if self.next_emit in self._pending:
                out.append(self._pending.pop(self.next_emit))
                self.next_emit += 1
            elif self.next_emit in self._dropped:
                self._dropped.discard(self.next_emit)
                self.next_emit += 1
            elif self._pending:
                # results are waiting behind a step that has not come back
                now = time.monotonic()
                if self._stalled_since is None:
                    self._stalled_since = now
                # a step queued in a worker that has died since (and whose results have been collected: poll() reads every
                # queue first) cannot come back: no point in waiting lost_after_s for it
                owner = self._owner.get(self.next_emit)
                orphan = owner is not None and not self._alive(owner) and owner.result_queue.empty()
                job = self._jobs.get(self.next_emit)
                if orphan and job is not None and job[3] < 3 and getattr(owner, "host_rings", None) is not None:
                    orphan = False   # (a host step: rescue_orphans() will queue it again as soon as a living worker has a free slot)
                if force or orphan or now - self._stalled_since > self.lost_after_s:
                    nxt = min(self._pending)  # lost downstream: do not stall the stream (poll() keeps every pending step >= next_emit)
                    gone = [s for s in range(self.next_emit, nxt) if s not in self._dropped]
                    self.lost_total += len(gone)
                    for g in gone:
                        self._jobs.pop(g, None)
                        self._owner.pop(g, None)   # (ring slots of a lost step stay taken: its worker may still write the result; they
                                                    #  come back with a late result, or go with the worker)
                    self._dropped.difference_update(range(self.next_emit, nxt))
                    print(f"StreamDispatcher: step(s) {gone} never came back, skipped", file=sys.stderr)
                    self.next_emit = nxt
                    self._stalled_since = None
                else:
                    break
            else:
                break
        if out or not self._pending:
            self._stalled_since = None
        for e in out:
            if e.profiler is not None and "upscaler.upscale" in e.profiler.data and e.frames is not None:
                e.profiler.set("upscaler.upscale.per_frame_ms", e.profiler.data["upscaler.upscale"] / len(e.frames) * 1000)
            if self.on_result is not None:
                self.on_result(e)
        return out

    def _collect(self) -> bool:
        """Read every service's result queue into ``_pending``; host results become views of the worker's output ring."""
        got_any = False
        for k, svc in enumerate(self.services):
            while True:
                try:
                    e = svc.result_queue.get_nowait()
                except queue.Empty:
                    break
                got_any = True
                held = self._slots.pop(e.step, None)
                self._jobs.pop(e.step, None)
                if isinstance(e.frames, HostFrames):
                    e.frames = svc.host_rings[1].view(e.frames.out_slot, e.frames.shape)
                    if held is not None:
                        self._pool(svc).give_in(held[1])                    # the input slot is free once the result exists
                        self._slots[("view", e.step)] = (svc, held[2])      # the output slot stays taken while the view is out
                prof = getattr(e, "profiler", None)
                if prof is not None and "upscaler.input.peer_copies" in getattr(prof, "data", {}):
                    self._peer_copies[k] = int(prof.data["upscaler.input.peer_copies"])
                if e.step < self.next_emit:
                    # its step was already passed (declared lost after lost_after_s, or skipped): emitting it now would
                    # put a stale frame behind newer ones and rewind next_emit, so it is counted and dropped
                    self.late_total += 1
                    print(f"StreamDispatcher: result of step {e.step} arrived late (stream is at {self.next_emit}), discarded", file=sys.stderr)
                    view = self._slots.pop(("view", e.step), None)
                    if view is not None:
                        self._pool(view[0]).give_out(view[1])
                    continue
                self._pending[e.step] = e
                self._owner.pop(e.step, None)
        return got_any

    def poll(self, timeout: float = 0.0) -> List[UpscalerQueueEntry]:
        """Collect finished jobs from every service and return those that can be emitted in order.  Frames that came back through a
        worker's host ring are views of it: valid until the next call of ``poll()``."""
        self._release_lent()
        deadline = time.monotonic() + timeout
        while True:
            got_any = self._collect()
            self.rescue_orphans()
            ready = self._emit_ready(force=len(self._pending) > self.max_reorder)
            if ready or time.monotonic() >= deadline:
                for e in ready:
                    view = self._slots.pop(("view", e.step), None)
                    if view is not None:
                        self._lent.append(view)
                        self._lent_steps.add(e.step)
                return ready
            if not got_any:
                time.sleep(0.001)

    def drain(self, expected_steps: Sequence[int], timeout: float = 60.0) -> List[UpscalerQueueEntry]:
        out, want = [], set(expected_steps)
        deadline = time.monotonic() + timeout
        while want and time.monotonic() < deadline:
            for e in self.poll(timeout=0.05):
                if e.step in self._lent_steps:
                    e.frames = e.frames.clone()   # (a ring view would not survive the next poll(): drain() collects over many)
                out.append(e)
                want.discard(e.step)
        return out

    def report(self) -> dict:
        return {"frame_step": self.frame_step, "dropped": self.dropped_total, "lost": self.lost_total, "late": self.late_total,
                "pending": len(self._pending), "rerouted": self.rerouted_total, "host_jobs": self.host_jobs_total,
                "host_fallback": self.host_fallback_total, "rescued": self.rescued_total, "peer_copies": [self._peer_copies.get(k, 0) for k in range(len(self.services))],
                "upscaler.inputq": [s.job_queue.qsize() for s in self.services]}
