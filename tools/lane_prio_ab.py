"""Dev tool (GPU box, dev library): priority of the context's lane stream (SS4K_LANE_PRIO: 0 normal, -1 high, 1 low), separate processes.
usage: SS4K_LIB=.../libss4k_hip_dev.so SS4K_LANE_PRIO=p python3 tools/lane_prio_ab.py [workload]"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
wl = sys.argv[1] if len(sys.argv) > 1 else "rrdbnet"
frames = B.synthetic_frames(4, (720, 1280), 1000).to("cuda")
s, _ = B.build_service(wl, 0)
def fps(reps=25):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): s.upscale(frames, wait=False)
    torch.cuda.synchronize(); return 4 * reps / (time.perf_counter() - t0)
for _ in range(10): s.upscale(frames, wait=False)
print(f"{wl} SS4K_LANE_PRIO={os.environ.get('SS4K_LANE_PRIO')}: {fps():.1f} {fps():.1f} {fps():.1f}", flush=True)
