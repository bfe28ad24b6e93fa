"""Small host utilities shared by the service mirror: the span profiler that rides in every queue
entry and the byte formatter the reference's pipelines print queue/bitrate sizes with."""

from .profiler import Profiler

__all__ = ["Profiler", "human_readable"]

_UNITS = ("B", "KB", "MB", "GB", "TB")

def human_readable(v, step=1024, unit=_UNITS):
    """``'1.5000MB'``-style rendering of a byte count: four decimals, the largest unit for which the
    value stays above one step (a value equal to a power of the step stays in the smaller unit),
    saturating at the last unit - the output format of the reference's ``src/util/__init__.py``."""
    if step <= 0:
        raise AssertionError("step must be positive")
    if v <= step:
        return f"{float(v):.4f}{unit[0]}"
    # largest k with v / step**k > 1 evaluated the way repeated division does (v > step at each stage)
    k = 0
    x = float(v)
    limit = len(unit) * 4 + 64  # repeated division terminates; bound it anyway
    while x > step and k < limit:
        x /= step
        k += 1
    return f"{x:.4f}{unit[min(k, len(unit) - 1)]}"
