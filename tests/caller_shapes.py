"""Interface-shaped stand-ins for the objects the reference's CALLERS put into the upscaler's queue.

The stream pipeline and the image server build their job record and their profiler from their own modules, not from this package.
What those types offer - and therefore all a drop-in may use - is: a record with the six fields ``frames, audio_segment, step,
elapsed, last_modified, profiler`` constructible by keyword, and a profiler with ``set / start / end`` and a ``data`` dict.  These
classes have exactly that and nothing else (no ``span``, no ``add``, no helper methods), so a test that pushes them through a worker
fails the moment the service leans on one of this package's own conveniences.  ``CallerPipeline`` is the wiring shape of the stream
caller: one object that owns the upscaler AND the next service, whose bound methods are the services' ``on_queue``.

Written for these tests; importable from a spawned worker (``tests`` is a package on ``sys.path``).
"""
from __future__ import annotations

import dataclasses
import time
from typing import Any

import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd.upscale.base_service import BaseService


@dataclasses.dataclass
class CallerEntry:
    frames: Any = None
    audio_segment: Any = None
    step: Any = 0
    elapsed: float = 0
    last_modified: float = 0
    profiler: Any = None


@dataclasses.dataclass
class SinkEntry:
    """What the stream caller hands to the service AFTER the upscaler (its own field names)."""
    frames: Any = None
    audio_segments: Any = None
    step: Any = 0
    profiler: Any = None


class CallerProfiler:
    """``set / start / end / data`` - the whole interface."""

    def __init__(self):
        self.data = {}
        self.opened = {}
        self.sums = {}

    def set(self, name, value):
        self.data[name] = value

    def start(self, name):
        self.opened[name] = time.time()

    def end(self, name):
        if name not in self.opened:
            return -1
        took = time.time() - self.opened.pop(name)
        tot, cnt = self.sums.get(name, (0.0, 0))
        self.sums[name] = (tot + took, cnt + 1)
        self.data[name] = self.sums[name][0] / self.sums[name][1]
        return took


class SinkService(BaseService):
    """The service after the upscaler in the caller's pipeline (the streamer's place): closes ``'upscaler.output'`` like it does and
    reports what arrived - a checksum, not the frames - on its result queue."""

    def proc_job_recieved(self, job):
        job.profiler.end("upscaler.output")
        frames = job.frames.cpu() if isinstance(job.frames, torch.Tensor) else torch.as_tensor(job.frames)
        return {"step": job.step, "shape": tuple(frames.shape), "sum": int(frames.to(torch.int64).sum()), "frames": frames,
                "keys": sorted(job.profiler.data), "sink_pid": __import__("os").getpid()}


class CallerPipeline:
    """Owns two services and wires them with its own bound methods, as the stream caller does: upscaler.on_queue runs INSIDE the
    upscaler's worker and pushes into the sink service's queue from there."""

    def __init__(self, upscaler_cls, **upscaler_kwargs):
        self.forwarded = 0
        self.upscaler = upscaler_cls(on_queue=self.upscaler_on_queue, **upscaler_kwargs)
        self.sink = SinkService()

    def upscaler_on_queue(self, entry):
        entry.profiler.start("upscaler.output.queue")
        frames = entry.frames.detach().clone()
        entry.profiler.set("upscaler.output.frames.shape", str(tuple(frames.shape)))
        entry.profiler.end("upscaler.output.queue")
        self.forwarded += 1
        self.sink.push_job_nowait(SinkEntry(frames=frames, audio_segments=entry.audio_segment, step=entry.step, profiler=entry.profiler))

    def start(self):
        self.sink.start()
        self.upscaler.start()

    def stop(self):
        self.upscaler.stop()
        self.sink.stop()
