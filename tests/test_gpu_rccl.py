"""RCCL on the hardware a one-GPU box has: the weight broadcast is the path's only collective, and at N = 1 it is
skipped - so without these tests the `nccl` backend would initialise on MI355X for the first time on somebody's 8-GPU node.
A world-1 `nccl` group proves library load, communicator init and the broadcast kernel on gfx950: everything short of xGMI.
Each case runs in a child process (a hung communicator must not take the test session with it; at most one extra GPU process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_BCAST = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import sharkshark4k_amd
from sharkshark4k_amd import sharding, weights as W
rank, world, local = sharding.init_distributed("nccl", force_group=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
dev = torch.device("cuda", local)
flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))          # the 33 MB (fp16) / 67 MB (fp32 blob) RRDBNet weights
got = sharding.broadcast_weights(flat, flat.size, dev, force_collective=True)
assert got.dtype == np.float32 and got.shape == flat.shape and got.tobytes() == flat.tobytes(), "broadcast changed the blob"
t = torch.ones(1 << 20, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
assert float(t.sum()) == float(1 << 20)
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", flat.size)
"""


def _env(**kw):
    return dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **kw)


def test_rccl_world1_broadcast_of_rrdbnet_blob():
    r = subprocess.run([sys.executable, "-c", _BCAST], cwd=ROOT, env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29641"),
                       capture_output=True, text=True, timeout=420)
    assert r.returncode == 0 and "RCCL_OK 16703171" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_spawned_rank_with_rccl_group():
    """`python bench.py --gpus 1 --spawn`: the parent never touches the GPU, starts the rank as a child (WORLD_SIZE=1) and relays
    its JSON line; SS4K_FORCE_GROUP=1 makes that rank create its RCCL group and push the weights through dist.broadcast."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--spawn", "--steps", "2", "--warmup", "1",
                        "--no-also", "--no-cpu-baseline", "--no-roofline"], cwd=ROOT, env=_env(SS4K_FORCE_GROUP="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 24.0 and "RCCL" in line["config"]["parallelism"], line


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher and SS4K_BENCH_REHEARSE_ON_ONE_GPU=1: the parent starts two ranks, both use cuda:0 and meet
    over gloo.  Every line of the N > 1 path runs: the group, rank 0 generating the weights, the broadcast into rank 1's service worker,
    the barrier-bracketed timed region, the max over ranks, rank 0's one JSON line relayed by the parent."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], cwd=ROOT, env=_env(SS4K_BENCH_REHEARSE_ON_ONE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                     # stdout is the one JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "REHEARSAL" in line["config"]["parallelism"], line
    assert line["value"] > 24.0 and abs(line["value"] - 2 * line["config"]["fps_per_gpu"]) < 1e-6   # whole-job value = all ranks' frames / max time
    assert "roofline" in line and "also" not in line and "cpu_baseline" not in line                  # N > 1: the headline record only
