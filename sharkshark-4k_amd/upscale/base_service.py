"""Worker-process runtime behind every service of the boundary.

Contract kept from the reference's ``BaseService`` (``src/upscale/base_service.py:10-121``), because
the pipelines drive it directly:

* ``start()`` launches ONE daemon worker process; ``push_job(entry, timeout=10)`` /
  ``push_job_nowait(entry)`` feed it, ``get_result(timeout=10)`` reads results (or the worker calls
  ``on_queue(entry)`` instead, when set); ``wait_for_job_clear()``, ``stop()``, ``join(timeout=15)``;
* queues ``job_queue`` / ``result_queue`` (32 deep) and ``cmd_queue`` (4096), the ``'exit'`` command;
* worker hooks ``proc_init()``, ``proc_job_recieved(job) -> entry``, ``proc_cleanup()``;
* failure policy: with ``exit_on_error`` a worker exception (or a dead worker noticed by the client,
  ``ProcessDeadException``) prints the traceback and interrupts the whole process group; without it
  the exception propagates and the worker dies; a full result queue drops the result with a warning;
* the object itself is what gets pickled into the worker, minus its ``Process`` handle.

Differences that matter on ROCm: it blocks on the job queue instead of spinning, on exit it flushes the result queue's feeder
thread before it terminates so that the last results are not lost, and the START METHOD is chosen when ``start()`` runs
(``mp_start_method = None``): the reference forks (``mp.Process`` on Linux), and its callers rely on it - the stream pipeline's
``on_queue`` is a bound method of an object that owns two more services and their fork-context queues (``pipeline.py:29-58``), the
image server's ``on_queue`` reads a module global that only a forked child inherits (``image_pipeline.py:38-47,54-64``); neither
survives pickling into a spawned interpreter.  A HIP context, on the other hand, does not survive ``fork``.  Both callers start their
services BEFORE anything in the parent touches the GPU, so: a parent whose GPU runtime is still untouched forks (exactly the
reference's behaviour, nothing has to be picklable), a parent that already holds a HIP context (tests, ``bench.py``, the multi-GPU
node's launcher after a warm-up) spawns a fresh interpreter and the service object travels by pickle.  ``mp_start_method = 'spawn'``
/ ``'fork'`` pins the choice; forking a parent that has initialised the GPU is refused with an error instead of a hung child.  The
queues are created from the spawn context either way (its semaphores are named, so they work in forked and in spawned children alike).

Two additions the multi-GPU node (``node.py``) and the jobs-in-flight overlap (``hip_upscaler.py``) use, both inert by default:
``ready_event`` (a ``multiprocessing.Event`` the worker sets when ``proc_init`` has returned, so that a launcher can wait for G
workers without sending a job) and HELD RESULTS: a service whose ``proc_job_recieved`` only ENQUEUES device work may keep up to
``proc_deliver_lag()`` finished-looking results back (in job order) while it reads and enqueues the next jobs, so that a consumer
callback that blocks on a result (``.cpu()``) does not stall the device.  A held result leaves as soon as ``proc_result_ready(entry)``
says its device work is done (polled every ``_HELD_POLL_S`` while the job queue is idle, so an isolated request is never kept waiting
for a successor), or when more than the lag are held; ``proc_before_deliver(entry)`` runs right before a result leaves the worker.
The defaults (lag 0, always ready) are the reference's loop: every result leaves before the next job is read.
"""
from __future__ import annotations

import abc
import collections
import os
import queue
import sys
import signal
import time
import traceback

import torch.multiprocessing as mp

_JOB_DEPTH = 32
_RESULT_DEPTH = 32
_COMMAND_DEPTH = 4096
_EXIT = "exit"
_IDLE_WAIT_S = 0.001
_HELD_POLL_S = 0.0002   # job-queue wait while results are held back: bounds how long a finished result sits in the worker


class ProcessDeadException(Exception):
    """The worker process is gone (raised on the client side when ``exit_on_error`` is set)."""


def _interrupt_process_group(ex: BaseException) -> None:
    traceback.print_exc()
    print(ex, file=sys.stderr)
    os.killpg(os.getpgid(os.getpid()), signal.SIGINT)


def gpu_runtime_touched() -> bool:
    """Has this process initialised the HIP runtime (through torch, through this package's library, or because a profiler that
    preloads itself did it before ``main``)?  A child forked from such a process cannot use the GPU."""
    import torch
    if torch.cuda.is_initialized():
        return True
    capi = sys.modules.get(__name__.rsplit(".", 2)[0] + "._capi")
    if capi is not None and getattr(capi, "GPU_TOUCHED", False):
        return True
    preload = " ".join(os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
    return "rocprofiler" in preload or "roctracer" in preload


class BaseService(abc.ABC):
    on_queue = None
    exit_on_error = False
    #: multiprocessing start method: None = fork while this process has not touched the GPU (the reference's method), else spawn
    mp_start_method = None
    #: most results held back (in order) while later jobs are being enqueued; 0 = every result leaves before the next job is read
    deliver_lag = 0
    #: set by the worker once proc_init() has returned (None: nobody waits for it)
    ready_event = None

    def __init__(self) -> None:
        mpctx = mp.get_context("spawn")   # (named semaphores: usable from a forked and from a spawned worker)
        self.job_queue = mpctx.Queue(maxsize=_JOB_DEPTH)
        self.result_queue = mpctx.Queue(maxsize=_RESULT_DEPTH)
        self.cmd_queue = mpctx.Queue(maxsize=_COMMAND_DEPTH)
        # callers read `proc` (sentinel, is_alive, exitcode) after start(); start() replaces it when the method turns out to be fork
        self.proc = mpctx.Process(target=self.proc_pre_main, daemon=True)

    def __getstate__(self):
        # the service object is the worker's target: everything travels except the process handle
        return {k: v for k, v in self.__dict__.items() if k != "proc"}

    # ---------------------------------------------------------------- worker side
    def proc_init(self) -> None:
        """Runs once in the worker before the first job (load models, create device contexts)."""

    def proc_job_recieved(self, job):
        """One job in, one result entry out."""
        return job

    def proc_cleanup(self) -> None:
        """Runs once in the worker after the exit command."""

    def proc_before_deliver(self, entry) -> None:
        """Runs in the worker right before ``entry`` is handed to ``on_queue`` / the result queue."""

    def proc_deliver_lag(self) -> int:
        """How many results may stay held right now (asked after every job: a service can make it depend on the newest job)."""
        return self.deliver_lag

    def proc_result_ready(self, entry) -> bool:
        """Is the device work behind this held result finished (so that handing it over cannot block)?"""
        return True

    def proc_pre_main(self) -> None:
        self.proc_main()

    def _exit_requested(self) -> bool:
        asked = False
        try:
            while True:  # drain: later commands must not pile up behind an exit
                asked |= self.cmd_queue.get_nowait() == _EXIT
        except queue.Empty:
            return asked

    def _deliver(self, entry) -> None:
        self.proc_before_deliver(entry)
        if self.on_queue is not None:
            self.on_queue(entry)
            return
        try:
            self.result_queue.put_nowait(entry)
        except queue.Full:
            print(f"{type(self).__name__}: result queue is full, result of this job dropped (consumer too slow?)", file=sys.stderr)

    def proc_main(self) -> None:
        try:
            self.proc_init()
            if self.ready_event is not None:
                self.ready_event.set()
            held = collections.deque()
            while not self._exit_requested():
                try:
                    job = self.job_queue.get(timeout=_HELD_POLL_S if held else _IDLE_WAIT_S)
                    held.append(self.proc_job_recieved(job))
                except queue.Empty:
                    pass
                lag = self.proc_deliver_lag()
                # in job order: whatever exceeds the lag leaves now (its consumer may block on it - the later jobs are already on the
                # device), and the oldest results leave as soon as they are ready, successor or not
                while held and (len(held) > lag or self.proc_result_ready(held[0])):
                    self._deliver(held.popleft())
            while held:
                self._deliver(held.popleft())
            self.proc_cleanup()
            self.result_queue.close()
            self.result_queue.join_thread()  # results still in the feeder thread reach the client first
        except Exception as ex:  # noqa: BLE001 - the policy below decides
            if not self.exit_on_error:
                raise
            _interrupt_process_group(ex)
            return
        print(f"{type(self).__name__}: worker leaves on request", file=sys.stderr)
        os.kill(os.getpid(), signal.SIGTERM)  # daemon threads of the runtime must not keep it alive

    # ---------------------------------------------------------------- client side
    def start_method(self) -> str:
        """'fork' or 'spawn' for a worker started NOW from this process (see the module docstring)."""
        touched = gpu_runtime_touched()
        method = self.mp_start_method
        if method is None:
            return "spawn" if touched else "fork"
        if method == "fork" and touched:
            raise RuntimeError(f"{type(self).__name__}: mp_start_method='fork' but this process has already initialised the GPU runtime - "
                               "a forked child cannot use it; start the service before the first GPU call or leave mp_start_method=None")
        return method

    def start(self) -> None:
        method = self.start_method()
        from .. import apply_runtime_env
        apply_runtime_env()   # the worker's HIP runtime initialises in the child: it inherits this environment (setdefault, package __init__)
        if method != "spawn":
            self.proc = mp.get_context(method).Process(target=self.proc_pre_main, daemon=True)
        self.proc.start()

    def check_proc(self) -> None:
        if self.exit_on_error and not self.proc.is_alive():
            try:
                raise ProcessDeadException("process is dead!")
            except ProcessDeadException as ex:
                _interrupt_process_group(ex)

    def push_job(self, entry, timeout=10) -> None:
        self.check_proc()
        self.job_queue.put(entry, timeout=timeout)

    def push_job_nowait(self, entry) -> None:
        self.check_proc()
        self.job_queue.put_nowait(entry)

    def get_result(self, timeout=10):
        self.check_proc()
        return self.result_queue.get(timeout=timeout)

    def wait_for_job_clear(self) -> None:
        while not self.job_queue.empty():
            time.sleep(_IDLE_WAIT_S)

    def join(self, timeout=15):
        self.proc.join(timeout=timeout)
        return self.proc.exitcode

    def stop(self):
        self.cmd_queue.put(_EXIT)
        return self.join()
