"""Dev probe (GPU box): for a service's job-set 0 stream against N candidate streams - the library's pair test (SS4K_LANE_CHECK_LOG=1 prints
its numbers) next to the service's own test (six alternating one-frame jobs, both streams against one).
usage: SS4K_LANE_CHECK_LOG=1 python3 tools/stream_pair_probe.py [N] [K]     K: streams created and used before the service exists"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
frames = B.synthetic_frames(1, (720, 1280), 1000).to(dev)
keep = [torch.cuda.Stream() for _ in range(K)]
for st in keep:
    with torch.cuda.stream(st): torch.zeros(1, device=dev)
torch.cuda.synchronize()
svc, _ = B.build_service("rrdbnet", 0)
svc._streams_checked = True          # (this probe does the checking)
sets = [svc._job_set(k) for k in range(2)]
ups = [svc._get_upscaler(k) for k in range(2)]
cur = torch.cuda.current_stream(dev)
def run(seq, stream_of):
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    used = {stream_of[i] for i in seq}
    for st in used: st.wait_stream(cur)
    for i in seq:
        with torch.cuda.stream(stream_of[i]): ups[i](frames)
    for st in used: cur.wait_stream(st)
    e1.record(cur); e1.synchronize()
    return e0.elapsed_time(e1)
s0 = sets[0]["stream"]
run((0, 1), {0: s0, 1: s0}); run((0, 1), {0: s0, 1: s0})
for n in range(N):
    c = sets[1]["stream"] if n == 0 else torch.cuda.Stream(dev)
    ok = sets[1]["ctx"].streams_side_by_side(s0, c)
    seq = (0, 1) * 3
    serial = run(seq, {0: s0, 1: s0}); both = run(seq, {0: s0, 1: c})
    ok_null = sets[1]["ctx"].streams_side_by_side(cur, c)
    print(f"candidate {n} ({c.cuda_stream:#x}): pair test vs set 0's stream {'ok ' if ok else 'BAD'}, vs the NULL stream {'ok ' if ok_null else 'BAD'}; six jobs {both:.2f} ms against {serial:.2f} in order = {both / serial:.2f}", flush=True)
