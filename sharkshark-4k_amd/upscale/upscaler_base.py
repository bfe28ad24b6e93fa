"""The job record and the upscaler-service contract of the drop-in boundary.

What callers of the reference depend on (``src/upscale/upscaler_base.py:17-63``, used from
``src/sharkshark/pipeline.py:73-118`` and ``image_server/image_pipeline.py:280-351``):

* ``UpscalerQueueEntry`` - the record that crosses the process boundary in both directions, with
  exactly these field names: ``frames`` (uint8 NHWC tensor, on the device), ``audio_segment``
  (passed through untouched), ``step`` (job ordinal), ``elapsed`` (seconds spent in the worker),
  ``last_modified`` (wall time the result was produced) and ``profiler``;
* ``BaseUpscalerService.proc_job_recieved(job) -> entry`` - closes the producer's
  ``'recoder.output'`` span, times ``self.upscale(job.frames)`` as ``'upscaler.upscale'`` and opens
  ``'upscaler.output'`` for the consumer to close;
* class attributes ``lr_shape`` / ``output_shape`` / ``on_queue`` that the pipelines overwrite.
"""
from __future__ import annotations

import dataclasses
import time
from typing import Any, Optional, Tuple

import torch

from ..util.profiler import Profiler
from .base_service import BaseService


@dataclasses.dataclass
class UpscalerQueueEntry:
    frames: Optional[torch.Tensor] = None
    audio_segment: Any = None
    step: Any = 0          # the stream pipeline counts jobs (pipeline.py:98), the image server passes a sha1 string (image_pipeline.py:283)
    elapsed: float = 0
    last_modified: float = 0
    profiler: Optional[Profiler] = None


#: the six fields every caller's entry type has (reference: upscaler_base.py:17-24) - the only thing this package assumes about a job
ENTRY_FIELDS = ("frames", "audio_segment", "step", "elapsed", "last_modified", "profiler")


def answer(job, frames, elapsed: float, profiler):
    """The result record of ``job``: same step / audio, new frames and timing - an entry of the CALLER's own type when that type takes the
    six fields as keywords (the reference's dataclass does, and so does any stand-in shaped like it), else this package's entry.  Nothing
    but the six fields is read from ``job``: a caller that imported ``UpscalerQueueEntry`` from its own tree keeps working."""
    fields = dict(frames=frames, audio_segment=getattr(job, "audio_segment", None), step=getattr(job, "step", 0),
                  elapsed=elapsed, last_modified=time.time(), profiler=profiler)
    cls = type(job)
    if cls is not UpscalerQueueEntry:
        try:
            return cls(**fields)
        except TypeError:
            pass
    return UpscalerQueueEntry(**fields)


def record_span(prof, name: str, seconds: float) -> None:
    """Put a span that was timed elsewhere (inside the native library) under ``prof.data[name]``.  This package's profiler folds it into
    the running mean (``add``); a caller's own profiler offers ``set / start / end / data`` only (reference: src/util/profiler.py:3-26) and
    gets the value through ``set``."""
    add = getattr(prof, "add", None)
    if add is not None:
        add(name, seconds)
    else:
        prof.set(name, seconds)


class BaseUpscalerService(BaseService):
    profiler: Profiler
    lr_shape: Tuple[int, int] = (720, 1280)
    output_shape: Optional[Tuple[int, int]] = (1440, 2560)
    on_queue = None

    def upscale(self, frames: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("an upscaler service implements upscale(uint8 NHWC) -> uint8 NHWC")

    def proc_job_recieved(self, job):  # (sic) the reference's spelling
        # only what the reference's own objects offer is used on `job` and its profiler (upscaler_base.py:17-24, util/profiler.py:3-26:
        # the six fields; set / start / end / data): the callers build both from THEIR modules (pipeline.py:95-100, image_pipeline.py:280-287)
        prof = job.profiler if getattr(job, "profiler", None) is not None else Profiler()
        self.profiler = prof  # upscale() implementations add their own spans to the job's profiler
        arrived = time.time()
        prof.end("recoder.output")
        prof.start("upscaler.upscale")
        try:
            upscaled = self.upscale(job.frames)
        finally:
            prof.end("upscaler.upscale")
        elapsed = time.time() - arrived
        prof.start("upscaler.output")
        return answer(job, upscaled, elapsed, prof)
