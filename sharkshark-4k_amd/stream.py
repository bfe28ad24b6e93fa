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

    # pipeline.py:61-108
    def submit_batch(self, frames, audio_segment=None, profiler: Optional[Profiler] = None) -> List[int]:
        """Cut one recorder batch into jobs and fan them out; returns the steps actually queued."""
        assert self.small_batch_size != 0
        profiler = profiler or Profiler()
        njobs = math.ceil(len(frames) / self.small_batch_size)
        queued = []
        for i in range(njobs):
            profiler.start("recoder.output.entry")
            chunk = torch.as_tensor(frames[i * self.small_batch_size:(i + 1) * self.small_batch_size])
            audio = None
            if audio_segment is not None:
                per = len(audio_segment) // njobs
                audio = torch.as_tensor(audio_segment[i * per:(i + 1) * per])
            step = self.frame_step
            self.frame_step += 1
            entry = UpscalerQueueEntry(frames=chunk, audio_segment=audio, step=step, profiler=profiler)
            profiler.set("recoder.output.frames.shape", str(tuple(chunk.shape)))
            profiler.end("recoder.output.entry")
            svc = self._pick(step)
            try:
                if self.frame_skips:
                    svc.push_job_nowait(entry)
                else:
                    svc.push_job(entry)
                queued.append(step)
                self._owner[step] = svc
            except queue.Full:
                self._dropped.add(step)
                self.dropped_total += 1
                print("StreamDispatcher: upscaler queue full, job skipped", file=sys.stderr)
        return queued

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
                if force or orphan or now - self._stalled_since > self.lost_after_s:
                    nxt = min(self._pending)  # lost downstream: do not stall the stream (poll() keeps every pending step >= next_emit)
                    gone = [s for s in range(self.next_emit, nxt) if s not in self._dropped]
                    self.lost_total += len(gone)
                    for g in gone:
                        self._owner.pop(g, None)
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

    def poll(self, timeout: float = 0.0) -> List[UpscalerQueueEntry]:
        """Collect finished jobs from every service and return those that can be emitted in order."""
        deadline = time.monotonic() + timeout
        while True:
            got_any = False
            for svc in self.services:
                try:
                    e = svc.result_queue.get_nowait()
                    got_any = True
                    if e.step < self.next_emit:
                        # its step was already passed (declared lost after lost_after_s, or skipped): emitting it now would
                        # put a stale frame behind newer ones and rewind next_emit, so it is counted and dropped
                        self.late_total += 1
                        print(f"StreamDispatcher: result of step {e.step} arrived late (stream is at {self.next_emit}), discarded", file=sys.stderr)
                        continue
                    self._pending[e.step] = e
                    self._owner.pop(e.step, None)
                except queue.Empty:
                    pass
            ready = self._emit_ready(force=len(self._pending) > self.max_reorder)
            if ready or time.monotonic() >= deadline:
                return ready
            if not got_any:
                time.sleep(0.001)

    def drain(self, expected_steps: Sequence[int], timeout: float = 60.0) -> List[UpscalerQueueEntry]:
        out, want = [], set(expected_steps)
        deadline = time.monotonic() + timeout
        while want and time.monotonic() < deadline:
            for e in self.poll(timeout=0.05):
                out.append(e)
                want.discard(e.step)
        return out

    def report(self) -> dict:
        return {"frame_step": self.frame_step, "dropped": self.dropped_total, "lost": self.lost_total, "late": self.late_total,
                "pending": len(self._pending), "rerouted": self.rerouted_total,
                "upscaler.inputq": [s.job_queue.qsize() for s in self.services]}
