"""Host frames in, host frames out: per-worker rings of shared, pinned host memory.

The reference's data flow starts and ends in host memory: the recorder process holds numpy frames and moves each job to the device
itself (``src/sharkshark/pipeline.py:84-93``: ``torch.tensor(...).to(device)`` + CUDA IPC), the streamer process copies every result back
(``src/stream/streamer.py:92-98``: ``.to('cpu').numpy()``).  On one GPU that is merely slow - a pageable 44 MB read-back costs as much as
the job that produced it (``profiles/earlier/r05/r05_pcie_inclusive.txt``: 63.9 against 127.0 frames/s).  On a node of G GPUs it is also wrong-shaped:
the caller would have to open a context on every GPU to put job ``s`` into the HBM of GPU ``s % G``.  So the node takes HOST frames and each
worker moves its own:

* the launcher (``node.UpscalerNode``, no HIP context) creates, per worker and once, two rings of ``slots`` fixed-size slots - frames in,
  frames out - as anonymous shared memory (``memfd``: no ``/dev/shm`` size limit, inherited by a forked worker, passed as a descriptor
  to a spawned one);
* the worker pins both (``hipHostRegister``) in ``proc_init``; a job is a ``HostFrames`` descriptor (ring slot + shape, ~ 100 bytes through the
  queue instead of 11 MB of pickled frames); H2D runs on a copy stream INSIDE the worker into a staging tensor, the job's launches wait for
  it by event, D2H runs on a second copy stream into the result slot; two jobs are in flight (``BaseService``: held results), so the copies
  of job i +- 1 run under the kernels of job i;
* the dispatcher hands the consumer a zero-copy view of the result slot, valid until the next ``poll()``.

Slot accounting lives in the parent only (``stream.StreamDispatcher``): a job owns one input and one output slot from submit until its
result has been handed out (input) / until the next ``poll()`` (output).
"""
from __future__ import annotations

import dataclasses
import mmap
import os
from multiprocessing import reduction
from typing import Optional, Tuple

import numpy as np
import torch

_PAGE = 4096
#: the rings this process has mapped, by id: a ``RingView`` that arrives through a queue finds its memory here
_RINGS = {}


@dataclasses.dataclass
class HostFrames:
    """What travels in ``UpscalerQueueEntry.frames`` instead of a tensor: the job's frames sit in slot ``slot`` of the worker's input
    ring (``shape`` = (N, H, W, 3) uint8), its result goes to slot ``out_slot`` of the output ring; on the way back ``shape`` is the
    result's shape and ``result`` is set."""
    slot: int
    out_slot: int
    shape: Tuple[int, int, int, int]
    result: bool = False

    def __len__(self) -> int:   # (callers take len(entry.frames) for per-frame figures)
        return int(self.shape[0])


class HostRing:
    """``slots`` x ``slot_bytes`` of shared host memory, addressable as uint8 tensors / arrays; pinned on request (worker side)."""

    def __init__(self, slots: int, slot_bytes: int, name: str = "ss4k_ring"):
        import uuid
        self.uid = uuid.uuid4().hex
        self.slots = int(slots)
        self.slot_bytes = (int(slot_bytes) + _PAGE - 1) // _PAGE * _PAGE
        self.fd = os.memfd_create(name)
        os.ftruncate(self.fd, self.slots * self.slot_bytes)
        self._map()

    def _map(self) -> None:
        self._mm = mmap.mmap(self.fd, self.slots * self.slot_bytes)      # MAP_SHARED: one memory for every process that maps the fd
        self._t = torch.frombuffer(self._mm, dtype=torch.uint8)
        self._pinned = False
        _RINGS[self.uid] = self

    # a spawned worker gets the descriptor (duplicated into the child by multiprocessing) and maps it again; a forked one inherits the map
    def __getstate__(self):
        return {"uid": self.uid, "slots": self.slots, "slot_bytes": self.slot_bytes, "fd": reduction.DupFd(self.fd)}

    def __setstate__(self, state):
        self.uid, self.slots, self.slot_bytes = state["uid"], state["slots"], state["slot_bytes"]
        self.fd = state["fd"].detach()
        self._map()

    @property
    def nbytes(self) -> int:
        return self.slots * self.slot_bytes

    def fits(self, shape) -> bool:
        return int(np.prod(shape)) <= self.slot_bytes

    def view(self, slot: int, shape) -> torch.Tensor:
        """uint8 tensor of ``shape`` over slot ``slot`` (no copy)."""
        n = int(np.prod(shape))
        if not (0 <= slot < self.slots and n <= self.slot_bytes):
            raise ValueError(f"HostRing: slot {slot} / {n} bytes outside a ring of {self.slots} x {self.slot_bytes}")
        at = slot * self.slot_bytes
        return self._t[at:at + n].view(tuple(int(s) for s in shape))

    def write(self, slot: int, frames) -> Tuple[int, int, int, int]:
        """Copy ``frames`` (numpy array or CPU tensor, uint8 NHWC) into a slot; returns the shape."""
        if isinstance(frames, torch.Tensor):
            assert frames.dtype == torch.uint8 and not frames.is_cuda
            shape = tuple(frames.shape)
            self.view(slot, shape).copy_(frames)
        else:
            a = np.asarray(frames)
            assert a.dtype == np.uint8
            shape = tuple(a.shape)
            np.copyto(self.view(slot, shape).numpy(), a)
        return shape

    def pin(self) -> bool:
        """Worker side, after its HIP context exists: page-lock the ring so that copies to and from it are real asynchronous DMA."""
        if self._pinned:
            return True
        rc = torch.cuda.cudart().cudaHostRegister(self._t.data_ptr(), self.nbytes, 0)
        rc = int(getattr(rc, "value", rc))
        if rc != 0:
            raise RuntimeError(f"HostRing: hipHostRegister of {self.nbytes} bytes failed with code {rc}")
        self._pinned = True
        return True

    def unpin(self) -> None:
        if self._pinned:
            torch.cuda.cudart().cudaHostUnregister(self._t.data_ptr())
            self._pinned = False

    def close(self) -> None:
        try:
            self.unpin()
        except Exception:  # noqa: BLE001 - the context may be gone already
            pass
        self._t = None
        _RINGS.pop(self.uid, None)
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass   # (views of the ring are still alive somewhere: the mapping goes with the process)
        try:
            os.close(self.fd)
        except OSError:
            pass


def _ring_view(uid: str, slot: int, shape):
    ring = _RINGS.get(uid)
    if ring is None:
        raise RuntimeError("a result frame arrived as a view of a host ring this process has not mapped: rings are created by the service "
                           "object before start() and reach forked / spawned children of that process only (clone() the frames before "
                           "handing them to another process)")
    return ring.view(slot, shape)


class RingView:
    """What a worker puts into the RESULT QUEUE for a host result: ~ 100 bytes that unpickle, in a process that has the ring mapped, as an
    ordinary uint8 CPU tensor over the slot - no per-result shared-memory segment (creating one 11 MB ``torch.multiprocessing`` segment
    per result costs ~ 90 ms in a process that holds a HIP context: profiles/r06_latency.txt)."""

    def __init__(self, ring: HostRing, slot: int, shape):
        self.uid, self.slot, self.shape = ring.uid, int(slot), tuple(int(s) for s in shape)

    def __reduce__(self):
        return _ring_view, (self.uid, self.slot, self.shape)

    def __len__(self) -> int:
        return self.shape[0]


def make_rings(slots: int, in_bytes: int, out_bytes: int) -> Tuple[HostRing, HostRing]:
    return HostRing(slots, in_bytes, "ss4k_frames_in"), HostRing(slots, out_bytes, "ss4k_frames_out")


class SlotPool:
    """Parent-side free lists of one worker's ring pair."""

    def __init__(self, slots: int):
        self.free_in = list(range(slots))
        self.free_out = list(range(slots))

    def take(self) -> Optional[Tuple[int, int]]:
        if not self.free_in or not self.free_out:
            return None
        return self.free_in.pop(), self.free_out.pop()

    def give_in(self, s: int) -> None:
        self.free_in.append(s)

    def give_out(self, s: int) -> None:
        self.free_out.append(s)
