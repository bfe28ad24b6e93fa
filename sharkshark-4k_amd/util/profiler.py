"""Named wall-clock spans that travel inside every queue entry.

The service boundary hands a ``Profiler`` along with each ``UpscalerQueueEntry`` and callers read
running means out of ``profiler.data`` under keys such as ``'upscaler.upscale'`` (reference:
``src/util/profiler.py:3-26``, read at ``src/sharkshark/pipeline.py:140-149``).  This keeps that
contract - ``start(name)`` / ``end(name)`` / ``set(name, value)`` and ``data[name]`` = mean seconds
of all closed spans of that name, ``end`` of a span that was never opened returns -1 and records
nothing - and adds a context manager, merging of the profilers of sharded workers and a report.

Spans are host wall-clock (``time.time()``), as in the reference: on an asynchronous device queue
they time the *enqueue*, not the kernels.  Device timing lives in the C
library (``ss4k_prof_*``) and in ``bench.py``.
"""
from __future__ import annotations

import contextlib
import time
from typing import Dict, Iterable, Iterator, List


class Profiler:
    """Picklable: plain dicts only (it crosses process boundaries inside queue entries)."""

    __slots__ = ("data", "_open", "_acc")

    def __init__(self) -> None:
        self.data: Dict[str, object] = {}      # name -> mean seconds (spans) or any value (set)
        self._open: Dict[str, float] = {}      # name -> wall time the span was opened at
        self._acc: Dict[str, List[float]] = {}  # name -> [total seconds, closed spans]

    # pickling with __slots__
    def __getstate__(self):
        return {k: getattr(self, k) for k in self.__slots__}

    def __setstate__(self, state):
        for k in self.__slots__:
            setattr(self, k, state[k])

    # -- the reference's interface -------------------------------------------------------------
    def set(self, name: str, value) -> None:
        self.data[name] = value

    def start(self, name: str) -> None:
        self._open[name] = time.time()

    def end(self, name: str) -> float:
        opened = self._open.pop(name, None)
        if opened is None:
            return -1
        took = time.time() - opened
        acc = self._acc.setdefault(name, [0.0, 0])
        acc[0] += took
        acc[1] += 1
        self.data[name] = acc[0] / acc[1]
        return took

    # -- additions ------------------------------------------------------------------------------
    def add(self, name: str, seconds: float) -> None:
        """Record a span that was timed elsewhere (e.g. inside the native library)."""
        acc = self._acc.setdefault(name, [0.0, 0])
        acc[0] += seconds
        acc[1] += 1
        self.data[name] = acc[0] / acc[1]

    @contextlib.contextmanager
    def span(self, name: str) -> Iterator[None]:
        self.start(name)
        try:
            yield
        finally:
            self.end(name)

    def is_open(self, name: str) -> bool:
        return name in self._open

    def count(self, name: str) -> int:
        return int(self._acc.get(name, (0.0, 0))[1])

    def total(self, name: str) -> float:
        return float(self._acc.get(name, (0.0, 0))[0])

    def merge(self, others: Iterable["Profiler"]) -> "Profiler":
        """Fold the closed spans of other profilers (e.g. one per GPU worker) into this one."""
        for other in others:
            for name, (tot, cnt) in other._acc.items():
                acc = self._acc.setdefault(name, [0.0, 0])
                acc[0] += tot
                acc[1] += cnt
                self.data[name] = acc[0] / acc[1]
            for name, value in other.data.items():
                if name not in other._acc:
                    self.data.setdefault(name, value)
        return self

    def report(self) -> str:
        rows = [f"{name}: {1000 * self.total(name) / max(1, self.count(name)):.3f} ms x {self.count(name)}"
                for name in sorted(self._acc)]
        rows += [f"{name} = {value}" for name, value in sorted(self.data.items()) if name not in self._acc]
        return "\n".join(rows)
