"""MI355X-native drop-in for sharkshark-4k's ``src/upscale`` per-frame super-resolution path.

Python host code (this package) mirrors the reference's service interface
(``BaseService`` / ``BaseUpscalerService`` / ``FsrcnnUpscalerService``) and calls a C-ABI HIP
library (``csrc/`` -> ``libss4k_hip.so``, declared in ``include/ss4k.h``) for all arithmetic.
There is no CPU fallback: constructing a context without the library or a GPU raises.
"""
import os as _os
import sys as _sys

#: The two HIP runtime settings this build's published numbers were measured with.  They are read when the runtime INITIALISES in a
#: process, so they belong in the environment a worker process starts with - importing this package does NOT set them (an import-time
#: side effect would reach a process that imported torch and touched the GPU first only by luck):
#:  * ``BaseService.start()`` applies them (``setdefault``: an integrator's own value wins) before the worker process is created, so every
#:    service worker - forked or spawned, single or one of a node's G - runs with them; ``bench.py`` sets them at its top;
#:  * a process that calls the library IN-PROCESS (``_capi`` directly) sets them itself before its first GPU call, or accepts the defaults;
#:    ``runtime_env_state()`` says which of the two it has.
#: HIP_FORCE_DEV_KERNARG=1: kernel-argument blocks in device memory (a forward is 213-351 launches; + 0.9 % on the headline job,
#: profiles/earlier/r05/r05_kernarg_ab.txt).  GPU_MAX_HW_QUEUES=8: room for a worker's five side-by-side streams (profiles/earlier/r05/r05_hwq_ab.txt); streams
#: that share a hardware queue run in order and some queue PAIRS are slow (profiles/earlier/r05/r05_lane_queue.txt), which the library and the service
#: test for whatever this is set to (ss4k_stream_pair_check).
RUNTIME_ENV = {"HIP_FORCE_DEV_KERNARG": "1", "GPU_MAX_HW_QUEUES": "8"}


def apply_runtime_env(env=None):
    """``setdefault`` the two settings into ``env`` (default: this process's environment, which child processes inherit)."""
    env = _os.environ if env is None else env
    for k, v in RUNTIME_ENV.items():
        env.setdefault(k, v)
    return env


def runtime_env_state() -> dict:
    """{name: (value in this process's environment or None, recommended)} - for logs; a GPU runtime that is already initialised read
    whatever was there at that moment."""
    return {k: (_os.environ.get(k), v) for k, v in RUNTIME_ENV.items()}


__version__ = "0.1.0"
