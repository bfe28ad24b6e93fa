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
    step: int = 0
    elapsed: float = 0
    last_modified: float = 0
    profiler: Optional[Profiler] = None

    def answered_by(self, frames: torch.Tensor, elapsed: float) -> "UpscalerQueueEntry":
        """The result record of this job: same step / audio / profiler, new frames and timing."""
        return dataclasses.replace(self, frames=frames, elapsed=elapsed, last_modified=time.time())


class BaseUpscalerService(BaseService):
    profiler: Profiler
    lr_shape: Tuple[int, int] = (720, 1280)
    output_shape: Optional[Tuple[int, int]] = (1440, 2560)
    on_queue = None

    def upscale(self, frames: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("an upscaler service implements upscale(uint8 NHWC) -> uint8 NHWC")

    def proc_job_recieved(self, job: UpscalerQueueEntry) -> UpscalerQueueEntry:  # (sic) the reference's spelling
        prof = job.profiler if job.profiler is not None else Profiler()
        self.profiler = prof  # upscale() implementations add their own spans to the job's profiler
        arrived = time.time()
        prof.end("recoder.output")
        with prof.span("upscaler.upscale"):
            upscaled = self.upscale(job.frames)
        result = job.answered_by(upscaled, elapsed=time.time() - arrived)
        result.profiler = prof
        prof.start("upscaler.output")
        return result
