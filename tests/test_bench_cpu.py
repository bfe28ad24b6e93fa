"""CPU: bench.py's launcher and its timing / max-over-ranks / report logic at world 2 over gloo.

The GPU work is replaced by a stub step (a rank-dependent sleep); everything else is the code the
driver runs: ``run_timed`` (warm-up, barrier-bracketed timed region, all_reduce MAX), ``world_seen``
(``n_gpus`` = ranks of the process group) and ``spawn_ranks`` (``python bench.py --gpus N`` with no
launcher starts N fresh ranks and fails loudly, in every rank, on a box without GPUs).
"""
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    b = _bench()
    from sharkshark4k_amd import sharding
    r, w_env, _ = sharding.init_distributed("gloo")
    world = b.world_seen(w_env)
    calls = []
    per_step = 0.02 if r == 0 else 0.06  # the slow rank sets the job's time

    def step():
        calls.append(time.perf_counter())
        time.sleep(per_step)

    elapsed = b.run_timed(step, steps=5, warmup=2, world=world, sync=lambda: None, device=torch.device("cpu"))
    q.put((r, world, len(calls), elapsed))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_run_timed_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, n0, e0), (r1, w1, n1, e1) = res
    assert (w0, w1) == (2, 2)                  # n_gpus is what the process group has
    assert n0 == n1 == 7                       # W warm-up + exactly K timed steps on every rank
    assert e0 == e1                            # both ranks report the MAX over ranks
    assert 5 * 0.06 <= e0 < 5 * 0.06 + 0.5     # ... which is the slow rank's time, warm-up excluded


def test_bench_without_launcher_fails_loudly_in_every_rank():
    """`python bench.py --gpus 2` on a box without GPUs: the parent starts two ranks before touching
    a GPU, both fail with the no-GPU assertion, the parent reports both and exits non-zero without a
    JSON line (no silent single-rank run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "rank 0: bench.py needs a GPU" in r.stderr and "rank 1: bench.py needs a GPU" in r.stderr
    assert "rank 0 exited with code" in r.stderr and "rank 1 exited with code" in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())


def test_world_size_mismatch_is_rejected():
    """One rank started by hand with WORLD_SIZE=1 but --gpus 2 must not report a 2-GPU number."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_host_cpu_report():
    n, model = _bench().host_cpu()
    assert n == os.cpu_count() and isinstance(model, str) and model
    json.dumps({"host_cpu_count": n, "host_cpu_model": model})


def test_live_traffic_declines_cleanly_and_aggregates_what_a_profiler_wrote(tmp_path, monkeypatch):
    """`live_pmc_traffic` never costs the bench line: no rocprofv3 -> (None, reason); a run that is itself under a profiler -> (None, reason); a
    counter pass that fails -> (None, reason).  And the aggregation it hands the passes to (tools/pmc_traffic.py) turns rocprofv3's
    counter_collection.csv rows into bytes per step: FETCH_SIZE doubled, WRITE_SIZE as is, KB x 1024, only the 3x3 conv kernels."""
    b = _bench()
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, "exists", lambda p: False if p == "/opt/rocm/bin/rocprofv3" else real_exists(p))
    assert b.live_pmc_traffic(4) == (None, "rocprofv3 not found")
    monkeypatch.setattr(shutil, "which", lambda name: "/bin/false")
    monkeypatch.setenv("ROCPROFILER_SDK_TOOL", "1")
    assert b.live_pmc_traffic(4) == (None, "this run is itself under a profiler")
    monkeypatch.delenv("ROCPROFILER_SDK_TOOL")
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    for k in [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_"))]:
        monkeypatch.delenv(k)
    got, why = b.live_pmc_traffic(4, timeout_s=30)          # "/bin/false -- python3 bench.py ..." exits with 1
    assert got is None and "FETCH_SIZE pass exited with 1" in why
    # the aggregation on two synthetic passes: 2 forwards of 3 conv launches + a kernel that must not be counted
    for counter, d in (("FETCH_SIZE", tmp_path / "f"), ("WRITE_SIZE", tmp_path / "w")):
        (d / "host" / "1").mkdir(parents=True)
        rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
        for i, name in enumerate(["void ss4k::w16::conv3x3_w16_kernel<false, true>(ss4k::ConvArgs)", "void ss4k::dense::conv3x3_dense2_kernel<8, false>(ss4k::DenseArgs)",
                                  "ss4k::w16n::conv3x3_w16n_kernel(ss4k::ConvArgs)", "ss4k::k_lane_spin(unsigned int)"] * 2):
            rows.append(f'{i},"{name}",{counter},{1000.0 if counter == "FETCH_SIZE" else 300.0}')
        (d / "host" / "1" / "x_counter_collection.csv").write_text("\n".join(rows) + "\n")
    out = tmp_path / "t.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "f"), str(tmp_path / "w"), str(out), "2", "2", "4"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    import json
    t = json.load(open(out))
    assert t["launches_counted"] == 6 and t["steps"] == 2 and t["frames_per_launch"] == 2
    assert t["traffic_bytes_per_step"] == (2 * 1000.0 + 300.0) * 1024 * 6 / 2
    assert set(t["by_kernel"]) == {"conv3x3_w16_kernel", "conv3x3_dense2_kernel", "conv3x3_w16n_kernel"}


def test_packed_record_bank_enumeration_matches_the_kernel_comment():
    """tools/costing/fm_packed_banks.py: the LDS cycles fsrcnn.hip quotes for the fp32-grade mapping stage's packed-record reads (40 per unit
    of 16 pixels with hi | lo interleaved records and pieces 8 ks + q + 4 e; the split-row layout is worse) are what the enumeration gives."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "costing"))
    try:
        import fm_packed_banks as fb
    finally:
        sys.path.pop(0)
    natural = fb.assigns["8ks+q+4e"]
    assert fb.cycles(fb.layout_interleaved(fb.FM_RW * 48), natural) == 40
    assert fb.cycles(fb.layout_split(fb.FM_RW * 24), natural) > 40
