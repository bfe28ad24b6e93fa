"""Dev tool (GPU box): the headline job with HOST frames on both sides ("with and without H2D / D2H").
  resident   frames and results stay in HBM (what bench.py reports)
  naive      pageable host frames in, `.cpu()` out, one job at a time - the way the reference's stream caller hands frames over (pipeline.py:91-92,118)
  pipelined  pinned host buffers, H2D and D2H on their own streams, two jobs in flight
usage: python3 tools/pcie_inclusive.py [steps]"""
import importlib.util, os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("ss4k_bench", os.path.join(ROOT, "bench.py")); B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
svc, _ = B.build_service("rrdbnet", 0)
host = B.synthetic_frames(4, (720, 1280), 1000)
res = svc.upscale(host.to(dev)); torch.cuda.synchronize()
for _ in range(10): svc.upscale(host.to(dev), wait=False)
torch.cuda.synchronize()

def resident():
    d = host.to(dev); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): svc.upscale(d, wait=False)
    torch.cuda.synchronize(); return 4 * steps / (time.perf_counter() - t0)

def naive():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): o = svc.upscale(host).cpu()
    return 4 * steps / (time.perf_counter() - t0), o

def pipelined(depth=2):
    h_in = [host.clone().pin_memory() for _ in range(depth)]
    h_out = [torch.empty(res.shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
    d_in = [torch.empty_like(host, device=dev) for _ in range(depth)]
    s_in, s_out, cur = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
    out_done = [None] * depth
    def run(n):
        outs = [None] * depth
        for i in range(n):
            k = i % depth
            if out_done[k] is not None: out_done[k].synchronize()          # the host buffer pair k is free again
            with torch.cuda.stream(s_in):
                d_in[k].copy_(h_in[k], non_blocking=True); e_in = s_in.record_event()
            cur.wait_event(e_in)
            outs[k] = svc.upscale(d_in[k])                                  # ordered on the current stream
            e_up = cur.record_event()
            with torch.cuda.stream(s_out):
                s_out.wait_event(e_up); h_out[k].copy_(outs[k], non_blocking=True); out_done[k] = s_out.record_event()
            outs[k].record_stream(s_out)
        torch.cuda.synchronize()
    run(6); t0 = time.perf_counter(); run(steps)
    return 4 * steps / (time.perf_counter() - t0), h_out[(steps - 1) % depth]

r = resident(); n, on = naive(); p, op = pipelined()
print(f"RRDBNet x2 720p -> 1440p, 4-frame jobs, {steps} steps: resident {r:.1f} frames/s; host frames, naive (pageable in, .cpu() out, one job at a time) {n:.1f}; "
      f"host frames, pinned + copy streams + two jobs in flight {p:.1f}; results equal: {bool(torch.equal(on, res.cpu()) and torch.equal(op, res.cpu()))}")
