"""CPU, world_size 2 over gloo: frame sharding by step and the one-off weight broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import sharding, weights as W


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sharkshark4k_amd  # noqa: F401
    from sharkshark4k_amd import sharding as sh, weights as Wt
    r, w, _ = sh.init_distributed("gloo")
    table = Wt.fsrcnn_table(seed=5) if r == 0 else None
    flat = sh.broadcast_weights(Wt.flatten(table, Wt.fsrcnn_keys()) if r == 0 else None, 12809, torch.device("cpu"))
    # every rank "upscales" its own steps (x2 nearest stands in for the GPU call)
    steps = list(range(7))
    local = {}
    for s in sh.my_steps(steps, r, w):
        frame = torch.full((1, 2, 3, 3), s, dtype=torch.uint8)
        local[s] = frame.repeat_interleave(2, 1).repeat_interleave(2, 2)
    merged = sh.gather_step_results(local, w)
    ordered = sh.reorder_results([(s, merged[s]) for s in sorted(merged, reverse=True)])
    out_q.put((r, float(flat.sum()), sorted(local), [int(t.flatten()[0]) for t in ordered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_sum = float(W.flatten(W.fsrcnn_table(seed=5), W.fsrcnn_keys()).sum())
    assert res[0][1] == res[1][1] == want_sum  # bit-identical weights on both ranks
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]  # step % world
    assert res[0][3] == res[1][3] == list(range(7))  # fan-in ordered by step


def test_single_process_paths():
    assert sharding.owner_of(5, 4) == 1
    assert sharding.my_steps(range(6), 1, 3) == [1, 4]
    flat = np.arange(4, dtype=np.float32)
    assert np.array_equal(sharding.broadcast_weights(flat, 4, torch.device("cpu")), flat)
    assert sharding.reorder_results([(2, "c"), (0, "a"), (1, "b")]) == ["a", "b", "c"]
