"""MI355X-native drop-in for sharkshark-4k's ``src/upscale`` per-frame super-resolution path.

Python host code (this package) mirrors the reference's service interface
(``BaseService`` / ``BaseUpscalerService`` / ``FsrcnnUpscalerService``) and calls a C-ABI HIP
library (``csrc/`` -> ``libss4k_hip.so``, declared in ``include/ss4k.h``) for all arithmetic.
There is no CPU fallback: constructing a context without the library or a GPU raises.
"""
import os as _os

# Kernel arguments in device memory (a HIP runtime setting, read when the runtime is LOADED - which `import torch` does: a process that
# imports torch first must already have these two in its environment, as bench.py and every spawned service worker do): a network forward is 213-351 launches with ~ 300-byte argument blocks, and fetching them from host memory costs
# launch latency that the launch chains expose.  Headline job, one box, three interleaved processes each: 123.9 -> 125.0 frames/s (+ 0.9 %,
# profiles/r05_kernarg_ab.txt).  setdefault: an integrator's own setting wins; service workers inherit it through the environment.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Eight hardware queues per process instead of HIP's four.  A service worker uses up to five streams that should run side by side (the
# current stream, three job-set streams, the context's frame-lane stream); streams that share a hardware queue run in order, whatever the
# program says (profiles/NOTES_r05.md 3).  Same A/B form: headline 124.9-125.6 either way; the SRVGG job of a process that had used many
# streams lost its second launch chain in two of three runs at four queues (402 against 423 frames/s) and never at eight
# (profiles/r05_hwq_ab.txt).  More queues are not only more room: some PAIRS of them are slow side by side (profiles/r05_lane_queue.txt), so
# the library tests its lane stream (ss4k_ctx::lane_check) and the service its job-set streams (hip_upscaler._vetted_stream) whatever this is set to.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
