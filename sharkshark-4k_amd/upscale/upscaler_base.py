"""Queue entry type and the upscaler service contract.

Mirrors the reference's ``src/upscale/upscaler_base.py:17-63``: ``UpscalerQueueEntry`` fields and
``BaseUpscalerService.proc_job_recieved`` (profiler spans ``'recoder.output'`` (end),
``'upscaler.upscale'``, ``'upscaler.output'`` (start); result entry carries ``elapsed`` and
``last_modified``).
"""
from __future__ import annotations

import time
from dataclasses import dataclass

import torch

from ..util.profiler import Profiler
from .base_service import BaseService


@dataclass
class UpscalerQueueEntry:
    frames: torch.Tensor = None
    audio_segment: torch.Tensor = None
    step: int = 0
    elapsed: float = 0
    last_modified: float = 0
    profiler: Profiler = None


class BaseUpscalerService(BaseService):
    profiler: Profiler
    lr_shape = (720, 1280)
    output_shape = (1440, 2560)
    on_queue = None

    def __init__(self) -> None:
        super().__init__()

    def proc_job_recieved(self, job: UpscalerQueueEntry):
        self.profiler = job.profiler
        began = time.time()
        job.profiler.end("recoder.output")
        job.profiler.start("upscaler.upscale")
        frames_up = self.upscale(job.frames)
        job.profiler.end("upscaler.upscale")
        elapsed = time.time() - began
        job.profiler.start("upscaler.output")
        return UpscalerQueueEntry(frames=frames_up, step=job.step, audio_segment=job.audio_segment, elapsed=elapsed,
                                  last_modified=time.time(), profiler=job.profiler)

    def upscale(self, frames):
        raise NotImplementedError
