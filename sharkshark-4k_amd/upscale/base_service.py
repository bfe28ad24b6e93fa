"""Worker-process service runtime: the boundary type the upscaler sits behind.

Mirrors the reference's ``BaseService`` (``src/upscale/base_service.py:10-121``): three
``torch.multiprocessing`` queues (jobs 32, results 32, commands 4096), one daemon worker process
running ``proc_init`` then a poll loop that drains commands, takes one job, calls
``proc_job_recieved`` and hands the result to ``on_queue`` or the result queue; the ``'exit'``
command; ``ProcessDeadException`` / kill-process-group on errors when ``exit_on_error``.
"""
from __future__ import annotations

import abc
import os
import signal
import time
import traceback
from queue import Empty, Full

import torch.multiprocessing as mp


class ProcessDeadException(Exception):
    pass


class BaseService(metaclass=abc.ABCMeta):
    on_queue = None
    exit_on_error = False
    #: multiprocessing start method.  'spawn' is required once the parent has touched the GPU
    #: (HIP contexts do not survive fork); the reference relies on the platform default.
    mp_start_method = "spawn"

    def __init__(self) -> None:
        ctx = mp.get_context(self.mp_start_method)
        self.job_queue = ctx.Queue(maxsize=32)
        self.result_queue = ctx.Queue(maxsize=32)
        self.cmd_queue = ctx.Queue(maxsize=4096)
        self.proc = ctx.Process(target=self.proc_pre_main, daemon=True)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("proc", None)  # a Process handle cannot be pickled into its own child
        return state

    def start(self):
        self.proc.start()

    def proc_pre_main(self):
        self.proc_main()

    def _drain_commands(self) -> bool:
        want_exit = False
        while True:
            try:
                if self.cmd_queue.get_nowait() == "exit":
                    want_exit = True
            except Empty:
                return want_exit

    def proc_main(self):
        try:
            self.proc_init()
            while not self._drain_commands():
                try:
                    job = self.job_queue.get_nowait()
                except Empty:
                    time.sleep(0.001)
                    continue
                entry = self.proc_job_recieved(job)
                try:
                    if self.on_queue is not None:
                        self.on_queue(entry)
                    else:
                        self.result_queue.put_nowait(entry)
                except Full:
                    print("BaseService.proc_main: result queue is full; is the consumer fast enough?")
            self.proc_cleanup()
            print("BaseService.proc_main: exit requested")
            # results still in the queue's feeder thread must reach the parent before we go
            self.result_queue.close()
            self.result_queue.join_thread()
            os.kill(os.getpid(), signal.SIGTERM)
        except Exception as ex:
            if self.exit_on_error:
                traceback.print_exc()
                print(ex)
                os.killpg(os.getpgid(os.getpid()), signal.SIGINT)
            else:
                raise

    def check_proc(self):
        if not self.exit_on_error:
            return
        if not self.proc.is_alive():
            try:
                raise ProcessDeadException("process is dead!")
            except Exception as ex:
                traceback.print_exc()
                print(ex)
                os.killpg(os.getpgid(os.getpid()), signal.SIGINT)

    def push_job(self, entry, timeout=10):
        self.check_proc()
        self.job_queue.put(entry, timeout=timeout)

    def push_job_nowait(self, entry):
        self.check_proc()
        self.job_queue.put_nowait(entry)

    def get_result(self, timeout=10):
        self.check_proc()
        return self.result_queue.get(timeout=timeout)

    def join(self, timeout=15):
        self.proc.join(timeout=timeout)
        return self.proc.exitcode

    def wait_for_job_clear(self):
        while not self.job_queue.empty():
            time.sleep(0.001)

    def stop(self):
        self.cmd_queue.put("exit")
        self.join()

    # hooks ---------------------------------------------------------------------------------
    def proc_init(self):
        pass

    def proc_job_recieved(self, job):
        pass

    def proc_cleanup(self):
        pass
