"""Named wall-clock spans that travel inside every queue entry.

Same interface and key semantics as the reference's ``src/util/profiler.py:3-26`` (``start`` /
``end`` / ``set`` and the running mean in ``data[name]``) because ``UpscalerQueueEntry.profiler``
is part of the service boundary and callers read keys such as ``'upscaler.upscale'``.
"""
import time


class Profiler:
    def __init__(self) -> None:
        self.start_ticks = {}
        self.data = {}
        self.elapsed_ticks = {}

    def set(self, name, value):
        self.data[name] = value

    def start(self, name):
        self.start_ticks[name] = time.time()

    def end(self, name):
        began = self.start_ticks.pop(name, None)
        if began is None:
            return -1  # span was never started: the reference reports -1 and records nothing
        elapsed = time.time() - began
        total, count = self.elapsed_ticks.get(name, (0, 0))
        self.elapsed_ticks[name] = (total + elapsed, count + 1)
        self.data[name] = (total + elapsed) / (count + 1)
        return elapsed
