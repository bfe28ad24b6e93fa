"""Dev probe (GPU box): does the speed of a 4-frame job with two launch chains depend on WHICH stream (in creation order) the context's lane
stream is?  Creates K used streams first, then builds the service and times it on the NULL stream.
usage: python3 tools/lane_queue_probe.py K [workload]"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
K = int(sys.argv[1]); wl = sys.argv[2] if len(sys.argv) > 2 else "rrdbnet"
dev = torch.device("cuda", 0)
frames = B.synthetic_frames(4, (720, 1280), 1000).to(dev)
keep = [torch.cuda.Stream() for _ in range(K)]
for st in keep:
    with torch.cuda.stream(st): torch.zeros(1, device=dev)
torch.cuda.synchronize()
s, _ = B.build_service(wl, 0)
def fps(reps=15):
    for _ in range(8): s.upscale(frames, wait=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): s.upscale(frames, wait=False)
    torch.cuda.synchronize(); return 4 * reps / (time.perf_counter() - t0)
print(f"K={K} {wl} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: {fps():.1f} {fps():.1f}", flush=True)
