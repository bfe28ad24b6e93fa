"""Small host utilities shared by the service mirror (currently the span profiler that rides in every queue entry)."""
from .profiler import Profiler

__all__ = ["Profiler"]
