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

Differences that matter on ROCm: the worker is spawned (a HIP context does not survive ``fork``), it
blocks on the job queue instead of spinning, and on exit it flushes the result queue's feeder thread
before it terminates so that the last results are not lost.

Two additions the multi-GPU node (``node.py``) and the one-frame-job overlap (``hip_upscaler.py``) use, both inert by default:
``ready_event`` (a ``multiprocessing.Event`` the worker sets when ``proc_init`` has returned, so that a launcher can wait for G
workers without sending a job) and ``deliver_lag`` (results handed over - still in job order - that many jobs late while the queue
keeps coming, and at once when it runs dry; ``proc_before_deliver(entry)`` runs right before a result leaves the worker).
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


class ProcessDeadException(Exception):
    """The worker process is gone (raised on the client side when ``exit_on_error`` is set)."""


def _interrupt_process_group(ex: BaseException) -> None:
    traceback.print_exc()
    print(ex, file=sys.stderr)
    os.killpg(os.getpgid(os.getpid()), signal.SIGINT)


class BaseService(abc.ABC):
    on_queue = None
    exit_on_error = False
    #: multiprocessing start method ('spawn': the parent may already hold a HIP context)
    mp_start_method = "spawn"
    #: results held back (in order) while later jobs are being enqueued; 0 = every result leaves before the next job is read
    deliver_lag = 0
    #: set by the worker once proc_init() has returned (None: nobody waits for it)
    ready_event = None

    def __init__(self) -> None:
        mpctx = mp.get_context(self.mp_start_method)
        self.job_queue = mpctx.Queue(maxsize=_JOB_DEPTH)
        self.result_queue = mpctx.Queue(maxsize=_RESULT_DEPTH)
        self.cmd_queue = mpctx.Queue(maxsize=_COMMAND_DEPTH)
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
                    job = self.job_queue.get(timeout=_IDLE_WAIT_S)
                except queue.Empty:
                    while held:  # the queue ran dry: nothing left to overlap with
                        self._deliver(held.popleft())
                    continue
                held.append(self.proc_job_recieved(job))
                while len(held) > self.deliver_lag:
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
    def start(self) -> None:
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
