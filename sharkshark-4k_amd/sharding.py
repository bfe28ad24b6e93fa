"""Frame sharding across the GPUs of one node: one process per GPU, no data-path collective.

The reference runs one upscaler process on one GPU (``pipeline.py:20``); every frame (job) is
independent, so jobs are dealt round-robin by ``step``: job ``s`` goes to rank
``s % world``.  The only collective is one broadcast of the packed weight blob from rank 0 at
start-up (RCCL over xGMI with backend ``nccl``; ``gloo`` in the CPU tests), so every rank runs
bit-identical weights without touching the filesystem/network again.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None, force_group: bool = False) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group.

    A world of one needs no group (nothing is exchanged); ``force_group`` (or ``SS4K_FORCE_GROUP=1``) creates it anyway, so
    that library load, communicator init and the broadcast kernel of RCCL can be exercised on a one-GPU box."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force_group = force_group or os.environ.get("SS4K_FORCE_GROUP", "0") == "1"
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


@dataclasses.dataclass
class GroupSpec:
    """Where a service worker finds the other workers of its node: what a launcher exports as RANK / WORLD_SIZE / MASTER_*, carried
    in the (pickled) service object instead, because ``node.UpscalerNode`` spawns its workers itself.  ``backend=None``: ``nccl``
    (RCCL) when the worker owns a GPU, else ``gloo``; two workers that share one GPU must say ``gloo`` (RCCL refuses duplicate devices).
    ``keep``: leave the group up after the weight broadcast (``bench.py`` needs its barrier); a service worker tears it down, so that
    the workers of a node are independent processes from then on - one that dies later cannot stall the others' exit."""
    rank: int = 0
    world: int = 1
    master_addr: str = "127.0.0.1"
    master_port: int = 29533
    backend: Optional[str] = None
    keep: bool = False
    force: bool = False   # create the group even for world == 1 (exercises library load / communicator / broadcast on a one-GPU box)


def join_group(spec: Optional[GroupSpec], local_device: Optional[int] = None) -> Tuple[int, int]:
    """(rank, world) of this worker.  A process that already sits in a group (``bench.py`` under ``torch.distributed.run``) keeps it;
    otherwise the group described by ``spec`` is joined (nothing to join for ``spec is None`` or a world of one without ``force``)."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    if spec is None or (spec.world <= 1 and not spec.force):
        return 0, 1
    backend = spec.backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl" and local_device is not None:
        torch.cuda.set_device(local_device)
    dist.init_process_group(backend=backend, init_method=f"tcp://{spec.master_addr}:{spec.master_port}", rank=spec.rank, world_size=spec.world)
    return spec.rank, spec.world


def leave_group() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()   # nobody tears the store down under a rank that is still inside the broadcast
        dist.destroy_process_group()


def owner_of(step: int, world: int) -> int:
    return step % world


def my_steps(steps: Iterable[int], rank: int, world: int) -> List[int]:
    return [s for s in steps if owner_of(s, world) == rank]


def broadcast_weights(flat: Optional[np.ndarray], n_floats: int, device: torch.device, src: int = 0,
                      force_collective: bool = False) -> np.ndarray:
    """Rank ``src`` passes the flat fp32 state_dict blob, the others ``None``; all get a copy.

    With one rank the blob is returned as it is - unless ``force_collective`` (or ``SS4K_FORCE_GROUP=1``) asks for the
    broadcast to run anyway (needs an initialised group): the blob then takes the same device round trip through
    ``dist.broadcast`` that it takes at N > 1."""
    force_collective = force_collective or os.environ.get("SS4K_FORCE_GROUP", "0") == "1"
    have_group = dist.is_available() and dist.is_initialized()
    if not have_group or (dist.get_world_size() == 1 and not force_collective):
        assert flat is not None
        return np.ascontiguousarray(flat, dtype=np.float32)
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")   # (two workers sharing one GPU, or the CPU tests: the blob goes through host memory)
    t = torch.empty(n_floats, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        assert flat is not None and flat.size == n_floats
        t.copy_(torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32)))
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def reorder_results(results: Sequence[Tuple[int, object]]) -> List[object]:
    """Fan-in: results arrive per rank in any interleaving; the sink emits them ordered by step
    (the reference's streamer only warns on out-of-order steps, ``streamer.py:77-78``)."""
    return [r for _, r in sorted(results, key=lambda sr: sr[0])]


def gather_step_results(local: Dict[int, torch.Tensor], world: int) -> Dict[int, torch.Tensor]:
    """Debug/test helper: collect every rank's {step: tensor} on all ranks (object collective;
    NOT on the timed path — production sinks read each rank's queue directly)."""
    if world == 1:
        return dict(local)
    payload = {s: t.cpu() for s, t in local.items()}
    out: List[Optional[dict]] = [None] * world
    dist.all_gather_object(out, payload)
    merged: Dict[int, torch.Tensor] = {}
    for d in out:
        merged.update(d)
    return merged
