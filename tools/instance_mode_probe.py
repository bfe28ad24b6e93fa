"""Dev probe (GPU box) of NOTES_r05 section 10's open observation: bench.py's `pipeline` leg ran 6-8 % low whenever the one-chain per-kernel pass had run
before it.  Replays that order with bench.py's own functions and drops one ingredient at a time.
usage: python3 tools/instance_mode_probe.py <variant>   variants: full | no_svc1_prof | no_svc1 | no_fsrcnn | no_main_prof | svc1_two_chains"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
from sharkshark4k_amd import _capi
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda", 0)
frames = B.synthetic_frames(4, (720, 1280), 1000).to(dev)
def fps(s, x, reps=15):
    for _ in range(8): s.upscale(x, wait=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): s.upscale(x, wait=False)
    torch.cuda.synchronize(); return x.shape[0] * reps / (time.perf_counter() - t0)
main, _ = B.build_service("rrdbnet", 0)
print(f"[{variant}] main {fps(main, frames):.1f}", end=" ", flush=True)
if variant != "no_main_prof": B.conv_roofline(main, frames, by_kernel=True)
if variant != "no_svc1":
    svc1, _ = B.build_service("rrdbnet", 0, flags=0 if variant == "svc1_two_chains" else _capi.MODEL_ONE_CHAIN)
    for _ in range(2): svc1.upscale(frames, wait=False)
    if variant != "no_svc1_prof": B.conv_roofline(svc1, frames, psteps=2, by_kernel=True)
    else: torch.cuda.synchronize()
    del svc1; torch.cuda.empty_cache()
if variant != "no_fsrcnn":
    for wl in ("fsrcnn", "fsrcnn_f16"):
        s2, _ = B.build_service(wl, 0); fps(s2, frames, 20); B.fsrcnn_stage_rooflines(s2, frames, half=wl == "fsrcnn_f16"); del s2; torch.cuda.empty_cache()
if variant.startswith("dummy"):   # shift the stream -> hardware queue mapping by creating (and using) N more streams first
    keep = [torch.cuda.Stream() for _ in range(int(variant[5:] or 1))]
    for st in keep:
        with torch.cuda.stream(st): torch.zeros(1, device=dev)
    torch.cuda.synchronize()
p, _ = B.build_service("pipeline", 0)
print(f"pipeline {fps(p, frames):.1f}", flush=True)
