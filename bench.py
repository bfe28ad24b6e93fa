#!/usr/bin/env python3
"""Headline benchmark: upscaled frames/s at 720p -> 1440p x2 on MI355X.

One "step" = one call of the frame-in/frame-out hot path (``ss4k_upscale_frames``: uint8 NHWC
frames resident in HBM -> uint8 NHWC upscaled frames in HBM) over one batch of synthetic frames.
Default workload (BASELINE.json configs[2], the one the >=24 fps target is quoted on):
RealESRGAN RRDBNet x2 (23 blocks), fp16 storage / fp32 accumulate.  Other BASELINE configs via
--workload {fsrcnn,rrdbnet,pipeline,srvgg}.

Every rank IS the product's service worker: ``HipUpscalerService(device=local_rank, ...).proc_init()`` joins the process group the launcher
set up, ONLY RANK 0 generates / repacks the weights and ``sharding.broadcast_weights`` (RCCL) hands the blobs to the others - the very code
``node.UpscalerNode``'s spawned workers run - and a step is ``service.upscale(frames)``.
Multi-GPU: one rank per GPU, frames sharded one-per-GPU with no data-path collective (weak scaling:
every rank runs the same per-GPU batch); the weight blobs are broadcast once from rank 0 over RCCL.
Launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or directly as ``python bench.py
--gpus N``: with WORLD_SIZE unset the parent starts the N ranks itself as fresh child processes
BEFORE it touches any GPU, waits for them and relays rank 0's JSON line.  Prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import math
import json
import os
import sys
import time

# HIP runtime settings of the package (sharkshark-4k_amd/__init__.py RUNTIME_ENV: kernel arguments in device memory, eight hardware queues).
# They belong in the environment a process has BEFORE its HIP runtime comes up; importing the package does not set them (BaseService.start()
# does, for the workers it creates).  This process is its own worker: set here, ahead of `import torch`
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd import _capi, sharding, weights as W  # noqa: E402
from sharkshark4k_amd.upscale import model as factory  # noqa: E402
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService  # noqa: E402

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16
F32_VECTOR_PEAK_TFLOPS = 157.3
PMC_TRAFFIC_FILE = "conv3x3_pmc_traffic_current.json"  # written by tools/collect_profiles.sh next to its per-round copy (tools/pmc_traffic.py)
# algorithmic bytes = uint8 frame in (2 764 800) + uint8 frame out (11 059 200) per frame, + the fp16 weights
# (33.4 MB) once per step: 88.7 MB for a 4-frame step
ALGORITHMIC_BYTES_PER_FRAME, ALGORITHMIC_WEIGHT_BYTES = 2_764_800 + 11_059_200, 33.4e6
HBM_SPEC_GBS, HBM_ACHIEVABLE_GBS = 8000.0, 6290.0  # MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured (float4 copy)
LAUNCHES_PER_FRAME = 213  # 23 blocks x 3 RDBs x (2 fused pairs + conv5) + conv_first, conv_body, conv_up1, conv_up2, conv_hr, conv_last

WORKLOADS = {
    "rrdbnet": "RealESRGAN RRDBNet x2 (23 blocks) 720p->1440p fp16 [BASELINE configs[2]]",
    "fsrcnn_f16": "FSRCNN x2 720p->1440p in the reference engine's precision: fp16 operands and intermediates, fp32 accumulation; all three stages on fp16 MFMA (dtype f16; PSNR against the fp32 CPU forward reported) [BASELINE configs[1] at TensorRT-fp16 precision]",
    "fsrcnn": "FSRCNN x2 720p->1440p, fp32 tensors; all three stages on fp16 MFMA with hi/lo-split operands and fp32 accumulation (fp32-grade: ~1e-6 of the exact kernels) [BASELINE configs[1]]",
    "pipeline": "BSVD denoise + RealESRGAN RRDBNet x2 720p->1440p fp16, per-frame path [BASELINE configs[3]]",
    "srvgg": "SRVGGNetCompact realesr-general-x4v3 x4 + bicubic to 1440p fp16 (the reference's shipped default)",
    "rrdbnet_x4": "RealESRGAN RRDBNet x4 (23 blocks) 1080p->4320x7680->bicubic 2160x3840 fp16 [BASELINE configs[4], per GPU]",
}


# workload -> (HipUpscalerService keyword arguments, output_shape, algorithmic FLOPs per input pixel).  Synthetic weights = the generated
# tables of weights.py (seed 0): there is no network on the GPU boxes to fetch checkpoints from.
SERVICE_OF = {
    "fsrcnn": (dict(upscaler_model="fsrcnn", scale=2, denoising=False, fsrcnn_dtype="f32"), None, 74784.0),
    "fsrcnn_f16": (dict(upscaler_model="fsrcnn", scale=2, denoising=False, fsrcnn_dtype="f16"), None, 74784.0),
    "rrdbnet": (dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False), None, 8.263e12 / (720 * 1280)),
    "pipeline": (dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=True, single_mode=True), None,
                 8.263e12 / (720 * 1280) + 590256.0),
    "rrdbnet_x4": (dict(upscaler_model="realesrgan", model_name="RealESRGAN_x4plus", denoising=False), (2160, 3840), 74.35e12 / (1080 * 1920)),
    "srvgg": (dict(upscaler_model="realesrgan", model_name="realesr-general-x4v3", denoise_rate=0.5, denoising=False), (1440, 2560), 2.0 * 1_209_024),
}


def build_service(workload, local, lr_shape=(720, 1280), flags=0, overlap_jobs=True):
    """-> (service with proc_init() done IN THIS PROCESS - the rank is the worker -, algorithmic FLOPs per frame of its networks).
    proc_init joins the rank's process group if there is one: rank 0 builds the weight blobs, the others receive them (RCCL)."""
    if workload not in SERVICE_OF:
        raise SystemExit(f"unknown workload {workload}")
    kw, out_shape, flop_px = SERVICE_OF[workload]
    svc = HipUpscalerService(device=local, weights="synthetic", seed=0, dtype="f16", lr_shape=lr_shape, model_flags=flags, overlap_jobs=overlap_jobs, **kw)
    svc.output_shape = out_shape
    svc.proc_init()
    return svc, flop_px * lr_shape[0] * lr_shape[1]


def live_pmc_traffic(batch, timeout_s=180):
    """roofline.traffic of THIS binary on THIS box: two child runs of this command under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: separate
    passes, with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes), aggregated by tools/pmc_traffic.py exactly like the
    committed collection (tools/collect_profiles.sh).  -> (dict of tools/pmc_traffic.py, None) or (None, why not).  The children are ordinary
    child processes of this one (nothing is exec'd over a process that holds the GPU); each is killed as a group if it overruns."""
    import shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    tmp = tempfile.mkdtemp(prefix="ss4k_pmc_", dir="/tmp")
    # two launch chains forced (SS4K_LANES=2) so that every conv launch carries batch / 2 frames; the child runs 7 settle calls + 1 warm-up + 2 steps
    env = dict(os.environ, TMPDIR="/tmp", SS4K_LANES="2")
    child = ["python3", os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", str(batch), "--no-cpu-baseline", "--no-roofline",
             "--no-also", "--no-by-kernel", "--no-live-traffic"]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, counter), "--"] + child
            p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)   # (exactly the group this call started)
                p.wait()
                return None, f"the {counter} pass did not finish in {timeout_s} s"
            if rc != 0:
                return None, f"the {counter} pass exited with {rc}"
        out = os.path.join(tmp, "traffic.json")
        r = subprocess.run(["python3", os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"), out,
                            str(max(1, batch // 2)), "10", str(batch)], cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=120)
        if r.returncode != 0:
            return None, "tools/pmc_traffic.py: " + r.stderr.strip().splitlines()[-1][:200]
        with open(out) as f:
            return json.load(f), None
    except Exception as e:  # noqa: BLE001 - a measurement that cannot be taken must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def synthetic_frames(batch, shape, seed):
    return torch.from_numpy(np.random.default_rng(seed).integers(0, 256, (batch, shape[0], shape[1], 3), dtype=np.uint8))


def cpu_baseline(workload, gpu_ctx, seconds_budget=14.0):
    """Oracle timed on this box's host cores on a bounded sample of the same workload (the largest
    crop of a 720p frame whose estimated time fits the budget, up to the whole frame); also the PSNR
    of the GPU production path against the oracle on that sample."""
    from oracle import nets as onets
    from oracle import service as osvc
    from tests.helpers import smooth_u8
    # many-core hosts oversubscribe small convs badly (256 threads: 285 s for a 96x160 crop): the thread count is the best of
    # a short sweep on a 180x320 crop (below)
    ncores = os.cpu_count() or 1
    torch.set_num_threads(min(16, ncores))
    if workload in ("fsrcnn", "fsrcnn_f16"):
        table = W.fsrcnn_table(0)
        def make(crop):
            osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=crop)
            sr = _capi.Model(gpu_ctx, _capi.make_desc(_capi.FSRCNN, _capi.F16 if workload == "fsrcnn_f16" else _capi.F32, scale=2),
                             W.flatten(table, W.fsrcnn_keys()))
            return osv, _capi.Upscaler(gpu_ctx, sr, crop, None, True, True, None, 1.0), sr
    else:
        table = W.rrdbnet_table(0, scale=2)
        flat = W.flatten(table, W.rrdbnet_keys(23))
        def make(crop):
            osv = osvc.OracleUpscaler(lambda x: onets.rrdbnet(x, table, 2, 23), upscaler_model="realesrgan", lr_shape=crop)
            sr = _capi.Model(gpu_ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
            return osv, _capi.Upscaler(gpu_ctx, sr, crop, None, True, False, None, 1.0), sr
    probe = (180, 320)
    osv, _, _ = make(probe)
    pf = torch.from_numpy(smooth_u8(7, (1, probe[0], probe[1], 3)))
    osv.upscale(pf)
    sweep = {}
    for nt in (16, 32, 64, 128):
        if nt > ncores and sweep:
            break
        torch.set_num_threads(min(nt, ncores))
        t0 = time.perf_counter(); osv.upscale(pf); sweep[torch.get_num_threads()] = time.perf_counter() - t0
        if sweep[torch.get_num_threads()] > 4.0 * min(sweep.values()):   # oversubscribed: more threads only get worse
            break
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    t_probe = sweep[best_nt]
    crop = probe
    for cand in ((720, 1280), (360, 640), (180, 320)):
        if t_probe * (cand[0] * cand[1]) / (probe[0] * probe[1]) <= seconds_budget:
            crop = cand
            break
    osv, up, keep = make(crop)
    frames = torch.from_numpy(smooth_u8(123, (1, crop[0], crop[1], 3)))
    t0 = time.perf_counter(); want = osv.upscale(frames); sec = time.perf_counter() - t0
    frac = (crop[0] * crop[1]) / (720 * 1280)
    got = up(frames.cuda()).cpu()
    mse = torch.mean((got.double() - want.double()) ** 2).item()
    psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    ncpu, cpu_model = host_cpu()
    return {"value": frac / sec, "unit": "frames/s (720p-frame equivalents)", "cores": torch.get_num_threads(),
            "host_cpu_count": ncpu, "host_cpu_model": cpu_model, "kind": "port",
            "thread_sweep_s_on_180x320_crop": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": f"oracle (PyTorch CPU fp32, {torch.get_num_threads()} threads = the fastest of a {sorted(sweep)} sweep) on one {crop[0]}x{crop[1]} frame crop "
                      f"= {frac:.4f} of a 720p frame: {sec:.2f} s"}, psnr


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh ranks (one per GPU) from a parent
    that has not initialised the GPU (no HIP call, no ``torch.cuda.is_available()``), relay rank 0's
    stdout and return the worst exit code.  Never re-execs a process that touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []  # rank 0's stdout is drained while it runs (a full pipe would block it)
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc, alive = 0, set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0:
                rc = rc or code
                print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        if rc and alive:  # one rank failed: the others would wait for it in the rendezvous / barrier
            time.sleep(2.0)
            for r in alive:
                if procs[r].poll() is None:
                    procs[r].kill()  # exactly the PIDs started above
        time.sleep(0.05)
    reader.join(timeout=30)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    return rc


def run_timed(step, steps, warmup, world, sync, device):
    """The timing contract: W untimed steps, then exactly K steps bracketed by device sync + barrier on
    both sides; the reported time is the MAX over ranks (one all_reduce of a scalar, outside the timed
    region).  ``sync`` is ``torch.cuda.synchronize`` on the GPU, a no-op in the CPU (gloo) test."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
    for _ in range(warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); barrier(); sync()
    elapsed = time.perf_counter() - t0
    run_timed.per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(every, t)      # (diagnostic: which rank was the slow one; outside the timed region)
        run_timed.per_rank = [float(x.item()) for x in every]
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def world_seen(world_env):
    """Ranks the process group really has (what ``n_gpus`` reports), checked against the launcher's claim."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        seen = torch.distributed.get_world_size()
        assert seen == world_env, f"process group has {seen} ranks, WORLD_SIZE says {world_env}"
        return seen
    return 1


def conv_roofline(svc, frames, psteps=3, by_kernel=False):
    """Dominant kernel timed live with HIP events on the launch stream (untimed extra steps)."""
    ctx = svc.ctx
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(psteps):
        svc.upscale(frames, wait=False)
    torch.cuda.synchronize()
    launches, ms, flops = ctx.prof_read()
    sec_ms = ctx.prof_read_section_ms()
    fams = ctx.prof_read_families() if by_kernel else None
    ctx.prof_enable(False)
    if launches <= 0 or ms <= 0 or sec_ms <= 0:
        return None
    # frame lanes: an even job may run as two concurrent launch chains (half of the frames each), so the per-launch
    # durations overlap.  The rate is therefore taken over the conv SECTION of each forward - wall time on the caller's
    # stream from its first conv launch to the end of its last, launch boundaries included: algorithmic FLOPs of all
    # launches / section time = FLOPs per launch / (average launch duration / launches in flight).  With one chain the
    # section time is the sum of the launch durations plus the boundaries between them.
    ach = flops / (sec_ms * 1e-3) / 1e12
    out = {"achieved": ach, "frac": ach / MFMA_F16_DENSE_PEAK_TFLOPS, "launches_per_step": launches / psteps,
           "avg_launch_us": 1000.0 * ms / launches, "algorithmic_gflop_per_launch": flops / launches / 1e9,
           "concurrent_launches": ms / sec_ms, "conv_ms_per_step": sec_ms / psteps}
    if fams:
        out["by_kernel"] = [{"kernel": name, "launches_per_step": n / psteps, "algorithmic_gflop_per_launch": fl / n / 1e9, "avg_launch_us": 1000.0 * t / n,
                             "share_of_kernel_time": t / ms, "tflops": fl / (t * 1e-3) / 1e12, "frac": fl / (t * 1e-3) / 1e12 / MFMA_F16_DENSE_PEAK_TFLOPS}
                            for name, n, t, fl in sorted(fams, key=lambda f: -f[2]) if n > 0 and t > 0]
    return out


def fsrcnn_stage_rooflines(svc, frames, psteps=3, half=False):
    """FSRCNN's three stages timed live (events around each stage on the launch stream), each against the unit that bounds it:
    all three run on the fp16 matrix cores with hi/lo-split operands - three MFMAs per product, so their algorithmic bound is the
    dense fp16 peak / 3 (the exact-fp32 vector-ALU head is kept for SS4K_MODEL_FS_EXACT).  half (dtype f16): every stage is
    one fp16 MFMA per product, against the dense fp16 peak."""
    ctx = svc.ctx
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(psteps):
        svc.upscale(frames, wait=False)
    torch.cuda.synchronize()
    stages = {}
    table = ((1, "head (5x5 conv 1->56 + 1x1 56->12, fp16 MFMA)", MFMA_F16_DENSE_PEAK_TFLOPS, "dense fp16 MFMA peak"),
             (2, "mapping (4 x conv3x3 12->12, fp16 MFMA)", MFMA_F16_DENSE_PEAK_TFLOPS, "dense fp16 MFMA peak"),
             (3, "tail (1x1 12->56 + 9x9 transposed conv, fp16 MFMA)", MFMA_F16_DENSE_PEAK_TFLOPS, "dense fp16 MFMA peak")) if half else (
                                  (1, "head (5x5 conv 1->56 + 1x1 56->12, fp16 MFMA, hi/lo split)", MFMA_F16_DENSE_PEAK_TFLOPS / 3, "dense fp16 MFMA peak / 3"),
                                   (2, "mapping (4 x conv3x3 12->12, fp16 MFMA, hi/lo split)", MFMA_F16_DENSE_PEAK_TFLOPS / 3, "dense fp16 MFMA peak / 3"),
                                   (3, "tail (1x1 12->56 + 9x9 transposed conv, fp16 MFMA, hi/lo split)", MFMA_F16_DENSE_PEAK_TFLOPS / 3, "dense fp16 MFMA peak / 3"))
    for kind, name, peak, unit in table:
        n, ms, fl = ctx.prof_read_kind(kind)
        if n > 0 and ms > 0:
            ach = fl / (ms * 1e-3) / 1e12
            stages[name] = {"ms_per_step": ms / psteps, "algorithmic_gflop_per_step": fl / psteps / 1e9, "achieved_tflops": ach,
                            "peak_tflops": peak, "peak_is": unit, "frac": ach / peak}
    ctx.prof_enable(False)
    return stages


def host_frames_leg(local, in_shape, batch, resident_fps, flops_per_frame, n_jobs=40, in_flight=3, resident_svc=None):
    from sharkshark4k_amd.node import UpscalerNode
    kw, _, _ = SERVICE_OF["rrdbnet"]
    node = UpscalerNode(devices=[local], fps=24 if batch >= 4 else batch, frame_skips=False, weights="synthetic", seed=0, dtype="f16", lr_shape=in_shape, **kw)
    node.start(timeout=600)
    try:
        job = node.dispatcher.small_batch_size
        host = synthetic_frames(job, in_shape, seed=2000).numpy()       # frames in pageable host memory, as a recorder holds them
        checksum = 0

        def pump(n):
            nonlocal checksum
            sent = got = 0
            while got < n:
                while sent < n and sent - got < in_flight:
                    sent += len(node.submit_batch(host))
                for e in node.poll(0.002):
                    assert not e.frames.is_cuda
                    checksum += int(e.frames[0, 0, 0, 0])               # (the consumer touches the host result)
                    got += 1
        pump(10)
        same = None
        if resident_svc is not None:   # one job's host result against the resident path's frames for the same input (byte for byte)
            got = None
            node.submit_batch(host)
            while got is None:
                for e in node.poll(0.01):
                    got = e.frames.clone()
            want = resident_svc.upscale(torch.from_numpy(host).to(resident_svc.torch_device)).cpu()
            same = bool(torch.equal(got, want))
        t1 = time.perf_counter(); pump(n_jobs); dt = time.perf_counter() - t1
        rep = node.report()
        fps = n_jobs * job / dt
        return {"workload": WORKLOADS["rrdbnet"] + f", HOST frames in and out through node.UpscalerNode (one spawned worker; pinned shared-memory rings, H2D / D2H on the "
                            f"worker's copy streams, {in_flight} jobs in flight): PCIe-inclusive, never the headline value",
                "frames_per_step": job, "fps": fps, "of_resident_value": fps / resident_fps, "net_tflops": flops_per_frame * fps / 1e12,
                "host_bytes_per_frame": in_shape[0] * in_shape[1] * 3 + 4 * in_shape[0] * in_shape[1] * 3, "identical_to_resident_path": same,
                "report": {k: rep[k] for k in ("host_jobs", "host_fallback", "lost", "dropped", "rescued")}}
    finally:
        node.stop()
        node.close()


def host_cpu():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4,
                    help="frames per step per GPU; 4 = the reference's stream-mode job size, small_batch_size = "
                         "min(4, fps) (src/sharkshark/pipeline.py:31,84), which is also what its README figure uses")
    ap.add_argument("--workload", default="rrdbnet", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short secondary-workload measurements")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed counter passes (profiles/) instead of two rocprofv3 --pmc child runs of this command")
    ap.add_argument("--no-by-kernel", action="store_true", help="skip the one-chain per-kernel pass of the roofline record (a second model: keeps a profiler trace to the headline job)")
    ap.add_argument("--spawn", action="store_true",
                    help="start the ranks as child processes even for --gpus 1 (exercises the parent / child relay; with "
                         "SS4K_FORCE_GROUP=1 the single rank also creates its RCCL group and broadcasts the weights through it)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        # no launcher: this process becomes the parent of N fresh ranks and never touches a GPU itself
        sys.exit(spawn_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--spawn"]))

    # stdout carries ONE JSON line: everything else this process, its libraries or its child processes (the service worker of the `also` leg)
    # might print to file descriptor 1 goes to stderr; the line itself is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # SS4K_BENCH_REHEARSE_ON_ONE_GPU=1 (tests only): the N ranks of a multi-GPU run all use cuda:0 and meet over gloo (RCCL refuses two
    # ranks on one device) - rehearses every line of the N > 1 path (group, rank-0 load, broadcast, barrier, max over ranks) on a one-GPU
    # box; the line's config says so and such a number is never a result
    rehearse = os.environ.get("SS4K_BENCH_REHEARSE_ON_ONE_GPU") == "1"
    rank, world_env, local = sharding.init_distributed("gloo" if rehearse else None)
    if rehearse:
        local = 0
    assert world_env == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch one rank per GPU"
    assert torch.cuda.is_available(), f"rank {rank}: bench.py needs a GPU (no CPU fallback exists)"
    assert local < torch.cuda.device_count(), f"rank {rank}: no GPU {local} on this node ({torch.cuda.device_count()} visible)"
    world = world_seen(world_env)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    in_shape = (1080, 1920) if args.workload == "rrdbnet_x4" else (720, 1280)
    # SS4K_BENCH_MODEL_FLAGS: SS4K_MODEL_* bits for the headline network (A/B of routes under the profiler)
    svc, flops_per_frame = build_service(args.workload, local, lr_shape=in_shape, flags=int(os.environ.get("SS4K_BENCH_MODEL_FLAGS", "0")))
    ctx = svc.ctx

    # every rank gets its own shard of the synthetic stream: frames rank, rank+world, ...
    frames = synthetic_frames(args.batch, in_shape, seed=1000 + rank).to(device)
    oh, ow = svc._get_upscaler().out_shape(args.batch, *in_shape)

    # the library settles its one / two launch chain choice for this job shape over the shape's first six forwards (models.cpp tune_step): done here,
    # ahead of the W warm-up steps, so that no timed step is a measurement step whatever W the caller passes
    for _ in range(7):
        svc.upscale(frames, wait=False)
    torch.cuda.synchronize()
    # a step = one job through the service's frame-in/frame-out call; results are ordered on the job set's stream (wait=False, what the
    # worker loop does) and the timed region ends with a device-wide synchronise
    elapsed = run_timed(lambda: svc.upscale(frames, wait=False), args.steps, args.warmup, world, torch.cuda.synchronize,
                        torch.device("cpu") if rehearse else device)
    total_frames = args.batch * args.steps * world
    fps = total_frames / elapsed

    result = {
        "metric": "upscaled frames/sec at 720p->1440p x2 (whole job)" if args.workload != "rrdbnet_x4"
                  else "upscaled frames/sec at 1080p->4K x4 (whole job)", "value": fps, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "per_rank_ms_per_step": [1000.0 * t / args.steps for t in run_timed.per_rank],   # each rank's own clock; ms_per_step is their max
        "dtype": "f32" if args.workload == "fsrcnn" else "f16", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload], "frames_per_step_per_gpu": args.batch,
                   "in": [in_shape[0], in_shape[1], 3], "out": [oh, ow, 3], "io": "uint8 NHWC resident in HBM",
                   "path": "HipUpscalerService.upscale -> ss4k_upscale_frames (every rank is a service worker: proc_init joins the group, rank 0 loads, RCCL broadcast)",
                   "parallelism": (f"REHEARSAL: {world} ranks sharing cuda:0 over gloo (SS4K_BENCH_REHEARSE_ON_ONE_GPU) - not a result" if rehearse else
                                   f"frame-sharded x{world}, weights broadcast once from rank 0 (RCCL)" if torch.distributed.is_initialized()
                                   else "one GPU, no process group (frames shard one-per-GPU at N > 1; the only collective is the weight broadcast)"),
                   "fps_per_gpu": fps / world, "net_tflops_per_gpu": flops_per_frame * fps / world / 1e12},
    }

    if rank == 0 and not args.no_roofline:
        rl = conv_roofline(svc, frames, by_kernel=True)
        if rl is not None:
            traffic, traffic_src, result_traffic_by_kernel = None, None, None
            pmc = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
            live, why_not = (None, "--no-live-traffic") if args.no_live_traffic else (None, "measured for the headline job on one GPU only") \
                if not (args.workload == "rrdbnet" and world == 1 and args.batch >= 2) else live_pmc_traffic(args.batch)
            if live is not None and live.get("launches_counted", 0) > 0:
                traffic = live["traffic_bytes_per_step"] / rl["launches_per_step"]
                traffic_src = ("live: two child runs of this command under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on this box, "
                               f"{live['launches_counted']} conv launches over {live['steps']} forwards, {live['frames_per_launch']} frames per launch (SS4K_LANES=2); " + live["correction"])
                result_traffic_by_kernel = {k: v["traffic_bytes_per_step"] for k, v in live.get("by_kernel", {}).items()}
            elif args.workload == "rrdbnet" and os.path.exists(pmc) and "traffic_bytes_per_step" in json.load(open(pmc)):
                # HBM-side bytes per launch cannot be read from inside the process: they come from the
                # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
                with open(pmc) as f:
                    pj = json.load(f)
                # bytes per step of all conv launches together (the counter passes ran the same job at pj["frames_per_step"] frames
                # per step); per launch = that / this run's launches per step
                bytes_step_pmc = pj["traffic_bytes_per_step"] * args.batch / pj["frames_per_step"]
                traffic = bytes_step_pmc / rl["launches_per_step"]
                traffic_src = f"profiles/{PMC_TRAFFIC_FILE} (" + pj["correction"] + f"; {pj['frames_per_launch']} frames per launch in the counter passes)" + \
                              f" [committed passes, not this run: {why_not}]"
                result_traffic_by_kernel = None
            mfma = {"achieved": rl["achieved"], "peak": MFMA_F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": rl["frac"]}
            hbm = None
            if traffic is not None:
                # the fabric-side bound at the MEASURED bytes (L2 <-> Infinity Cache / HBM; the counters include Infinity-Cache
                # hits, MI355X_MICROARCH.md HBM section): bytes per step / conv time of a step
                bytes_step = traffic * rl["launches_per_step"]
                gbs = bytes_step / (rl["conv_ms_per_step"] * 1e-3) / 1e9
                hbm = {"what": "diagnostic: measured L2 <-> fabric bytes over conv time (NOT an achievement figure)",
                       "achieved": gbs, "peak": HBM_SPEC_GBS, "unit": "GB/s", "frac": gbs / HBM_SPEC_GBS,
                       "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": gbs / HBM_ACHIEVABLE_GBS,
                       "bytes_per_step": bytes_step,
                       "flop_per_byte": rl["algorithmic_gflop_per_launch"] * 1e9 / traffic,
                       "ridge_flop_per_byte": MFMA_F16_DENSE_PEAK_TFLOPS * 1e12 / (HBM_ACHIEVABLE_GBS * 1e9),
                       "note": "measured fabric bytes (layer-by-layer dense blocks re-read x 5x, x1 4x, ... per RDB), not SURVEY 8(d)'s "
                               "algorithmic bytes; FETCH_SIZE counts Infinity-Cache hits, so the HBM share of these bytes is unknown"}
            # this path is MFMA-bound by arithmetic intensity (8.263 TFLOP against 22 MB of algorithmic bytes per
            # frame), so the figure of merit is the MFMA fraction at ALGORITHMIC FLOPs.  The fabric figures (measured L2 <-> fabric
            # bytes, FLOP per measured byte against the ridge) are the diagnosis of what holds it back - they go UP when a kernel
            # wastes more bytes - and live in the "fabric" sub-record, never in frac.
            alg_bytes_step = ALGORITHMIC_BYTES_PER_FRAME * args.batch + ALGORITHMIC_WEIGHT_BYTES
            if hbm is not None:
                if result_traffic_by_kernel:
                    hbm["bytes_per_step_by_kernel"] = result_traffic_by_kernel
                hbm["algorithmic_bytes_per_step"] = alg_bytes_step
                hbm["measured_over_algorithmic"] = hbm["bytes_per_step"] / alg_bytes_step
            result["roofline"] = {"bound": "mfma", "achieved": mfma["achieved"], "peak": mfma["peak"], "unit": mfma["unit"],
                                  "frac": mfma["frac"], "mfma": mfma, "fabric": hbm, "traffic": traffic,
                                  "algorithmic_bytes_per_step": alg_bytes_step,
                                  "traffic_unit": "bytes per launch, L2 <-> fabric (Infinity Cache / HBM), PMC", "traffic_source": traffic_src,
                                  # the kernel builds this run's conv launches were routed to (ss4k_prof_read_family), by share of kernel time
                                  "kernel": "3x3 implicit-GEMM conv, all launches of a step: " + " + ".join(k["kernel"].split(" (")[0] for k in rl.get("by_kernel", [])),
                                  "launches_per_step": rl["launches_per_step"],
                                  "avg_launch_us": rl["avg_launch_us"],
                                  "algorithmic_gflop_per_launch": rl["algorithmic_gflop_per_launch"],
                                  "concurrent_launches": rl["concurrent_launches"],
                                  "launches_per_frame_chain": LAUNCHES_PER_FRAME,
                                  "kernel_time_share_of_step": rl["conv_ms_per_step"] / (1000.0 * elapsed / args.steps)}
        elif args.workload in ("fsrcnn", "fsrcnn_f16"):
            # FSRCNN: three stages, each against the unit that bounds it (fsrcnn_stage_rooflines); the line's roofline is the
            # stage that takes the longest
            stages = fsrcnn_stage_rooflines(svc, frames, half=args.workload == "fsrcnn_f16")
            if stages:
                name, dom = max(stages.items(), key=lambda kv: kv[1]["ms_per_step"])
                result["roofline"] = {"bound": "mfma", "achieved": dom["achieved_tflops"], "peak": dom["peak_tflops"], "unit": "TFLOP/s",
                                      "frac": dom["frac"], "traffic": None, "kernel": "fsrcnn stage: " + name, "peak_is": dom["peak_is"],
                                      "stages": stages}
    if rank == 0 and world == 1 and not args.no_also and args.workload == "rrdbnet":
        # the other single-GPU BASELINE configs and the 1-frame (image-server / latency) job, measured the
        # same way (short, outside the headline timing); conv-based ones carry their own roofline fraction
        also = {}

        def timed(svc2, fr2, reps, settle=7):
            for _ in range(settle):   # (the library measures one / two launch chains over a shape's first six forwards: keep that out of the timing)
                svc2.upscale(fr2, wait=False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            for _ in range(reps):
                svc2.upscale(fr2, wait=False)
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        for name, wl, nb, shape, reps in (("fsrcnn", "fsrcnn", args.batch, (720, 1280), 200),      # (a step is 1.7 / 0.7 ms: enough of them to time, after
                                          ("fsrcnn_f16", "fsrcnn_f16", args.batch, (720, 1280), 400),  #  enough of them to be past the clock ramp: below)
                                          ("pipeline", "pipeline", args.batch, (720, 1280), 15),
                                          ("srvgg", "srvgg", args.batch, (720, 1280), 30),
                                          ("rrdbnet_n1", "rrdbnet", 1, (720, 1280), 40),
                                          ("rrdbnet_n1_one_set", "rrdbnet", 1, (720, 1280), 60),   # (0.55 s: a 0.18 s burst after the service build measured 108-116 by how long the chip had idled)
                                          ("rrdbnet_x4", "rrdbnet_x4", 1, (1080, 1920), 5)):
            # rrdbnet_n1: back-to-back one-frame jobs from ONE service (the image server's caller): consecutive jobs alternate over the
            # service's three job sets (hip_upscaler.py); rrdbnet_n1_one_set: the same with overlap_jobs=False (every job on one set / stream)
            if name == "rrdbnet_n1":
                svc2, fpf = svc, flops_per_frame
            else:
                svc2, fpf = build_service(wl, local, lr_shape=shape, overlap_jobs=name != "rrdbnet_n1_one_set")
            fr2 = frames[:nb] if shape == in_shape else synthetic_frames(nb, shape, seed=77).to(device)
            # FSRCNN: a service build idles the chip for ~ 1 s and a 0.1 s burst right after it runs at the clock of the ramp, not the one the
            # job holds (in-bench 5 590 against 5 690 frames/s stand-alone with 20 warm-up steps): 100 untimed steps first
            dt = timed(svc2, fr2, reps, settle=100 if wl.startswith("fsrcnn") else 7)
            also[name] = {"workload": WORKLOADS[wl] + (", back-to-back one-frame jobs from one service, alternating over its three job sets" if name == "rrdbnet_n1" else
                                                       ", one-frame jobs on one job set (overlap_jobs=False)" if name == "rrdbnet_n1_one_set" else ""),
                          "frames_per_step": nb, "fps": reps * nb / dt, "net_tflops": fpf * reps * nb / dt / 1e12}
            if name == "rrdbnet_n1":
                also[name]["job_sets"] = len(svc2._sets)
            elif not wl.startswith("fsrcnn"):
                # (one-frame legs: on ONE job set - with consecutive jobs alternating over the sets a context's conv sections share the chip with
                # the other sets' jobs, and the per-context rate is not the chip's)
                keep_overlap, svc2.overlap_jobs = svc2.overlap_jobs, svc2.overlap_jobs and nb > svc2.overlap_max_frames
                rl = conv_roofline(svc2, fr2, psteps=2)
                svc2.overlap_jobs = keep_overlap
                if rl is not None:
                    also[name]["conv_tflops"] = rl["achieved"]; also[name]["conv_frac_of_peak"] = rl["frac"]
                    also[name]["conv_launches_per_step"] = rl["launches_per_step"]
            else:
                # algorithmic fp32 FLOPs against the fp32 vector / matrix peak that bounds an exact-fp32 implementation (the
                # fp16-split stages are not bound by it: context, not a roofline fraction)
                if wl == "fsrcnn":
                    also[name]["frac_of_fp32_peak"] = also[name]["net_tflops"] / F32_VECTOR_PEAK_TFLOPS
                else:
                    # the fp16 mode's uint8 frames against the fp32-accurate mode's on the same job
                    svc3, _ = build_service("fsrcnn", local, lr_shape=shape)
                    out3, out2 = svc3.upscale(fr2), svc2.upscale(fr2)
                    torch.cuda.synchronize()
                    d = out2.to(torch.int16) - out3.to(torch.int16)
                    mse = float((d.float() ** 2).mean())
                    also[name]["u8_vs_f32_mode"] = {"psnr_db": None if mse == 0 else 10 * math.log10(255.0 ** 2 / mse),
                                                    "max_lsb": int(d.abs().max()), "frac_differing": float((d != 0).float().mean())}
                    del svc3, out3, out2
                also[name]["stages"] = fsrcnn_stage_rooflines(svc2, fr2, half=wl == "fsrcnn_f16")
            del fr2
            if name != "rrdbnet_n1":
                del svc2
            torch.cuda.empty_cache()
        # ... and the same one-frame jobs through a REAL worker process of the service (spawned child, job / result queues, CUDA-IPC
        # tensors, results held until ready / at most two while jobs alternate): what an integrator's image server gets from one GPU
        try:
            from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
            kw, _, _ = SERVICE_OF["rrdbnet"]
            wsvc = HipUpscalerService(device=local, weights="synthetic", seed=0, dtype="f16", lr_shape=in_shape, **kw)
            wsvc.start()
            fr1 = frames[:1].clone()
            n_jobs, depth = 60, 8
            def pump(n):
                sent = got = 0
                while got < n:
                    while sent < n and sent - got < depth:
                        wsvc.push_job(UpscalerQueueEntry(frames=fr1, step=sent), timeout=600)
                        sent += 1
                    wsvc.get_result(timeout=600)
                    got += 1
            pump(6)
            t1 = time.perf_counter(); pump(n_jobs); dt = time.perf_counter() - t1
            wsvc.stop()
            also["rrdbnet_n1_worker"] = {"workload": WORKLOADS["rrdbnet"] + f", one-frame jobs through a spawned service worker (queues + CUDA IPC, {depth} jobs in flight)",
                                         "frames_per_step": 1, "fps": n_jobs / dt, "net_tflops": flops_per_frame * n_jobs / dt / 1e12}
        except Exception as e:  # never lose the headline line to a secondary measurement
            also["rrdbnet_n1_worker"] = {"error": f"{type(e).__name__}: {e}"}
        # ... and HOST frames on both sides through the node (SURVEY 8(d) "with and without H2D/D2H"): UpscalerNode with one spawned worker on
        # this GPU, numpy frames in host memory -> the worker's pinned input ring -> H2D on the worker's copy stream -> the job -> D2H into
        # the pinned output ring -> a host view at the sink; 4-frame jobs, three in flight.  PCIe-inclusive: never the headline `value`.
        try:
            also["host_frames"] = host_frames_leg(local, in_shape, args.batch, fps, flops_per_frame, resident_svc=svc)
        except Exception as e:  # never lose the headline line to a secondary measurement
            also["host_frames"] = {"error": f"{type(e).__name__}: {e}"}
        result["also"] = also
    if rank == 0 and "roofline" in result and args.workload == "rrdbnet" and world == 1 and not args.no_by_kernel:
        # per kernel build, from a ONE-CHAIN run of the same job (SS4K_MODEL_ONE_CHAIN: with two launch chains in flight a launch's
        # duration includes the time it shares the chip with the other chain's launch - nobody should have to divide by 1.9).
        # Placed after every other measurement: before the library tested its lane stream (ss4k_stream_pair_check), a leg that ran AFTER
        # this pass could get the hardware queue that is slow beside the NULL stream's and measure 6-8 % low (profiles/NOTES_r05.md 10)
        try:
            svc1, _ = build_service("rrdbnet", local, lr_shape=in_shape, flags=_capi.MODEL_ONE_CHAIN)
            for _ in range(2):
                svc1.upscale(frames, wait=False)
            rl1 = conv_roofline(svc1, frames, psteps=2, by_kernel=True)
            result["roofline"]["by_kernel"] = {"what": "one launch chain (SS4K_MODEL_ONE_CHAIN), every launch alone on the chip: algorithmic GFLOP / average launch "
                                                       "duration (HIP events on the launch stream) against the dense fp16 MFMA peak",
                                               "frames_per_launch": args.batch, "conv_tflops": rl1["achieved"], "conv_frac_of_peak": rl1["frac"],
                                               "kernels": rl1.get("by_kernel", [])}
            del svc1
            torch.cuda.empty_cache()
        except Exception as e:  # never lose the headline line to a secondary measurement
            result["roofline"]["by_kernel"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, psnr = cpu_baseline(args.workload, _capi.Context(local))
        result["cpu_baseline"] = cb
        result["psnr_db_vs_cpu_ref"] = psnr
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
