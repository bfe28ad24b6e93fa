"""Dev tool (GPU box): FSRCNN x2 on 12 planes of 720p, per-stage ms (ss4k_prof_read_kind) in fp16 mode and fp32-grade mode, for the values of an
environment switch read per call / per process: python tools/fs_stage_ab.py SS4K_MH_NU 1 2   (one child process per value, interleaved)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import sharkshark4k_amd  # noqa
    from sharkshark4k_amd import _capi, weights as W
    from sharkshark4k_amd.upscale import model as factory
    ctx = _capi.Context(0)
    x = torch.rand(12, 1, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
    for dt in ("f16", "f32"):
        m = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2), dtype=dt)
        for _ in range(5): m(x)
        torch.cuda.synchronize()
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        ms = [ctx.prof_read_kind(k)[1] / 20 for k in (1, 2, 3)]
        ctx.prof_enable(False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): m(x)
        e1.record(); torch.cuda.synchronize()
        y = m(x[:3, :, :200, :300]).double()
        print(f"RES {dt}: head {ms[0]:.4f} map {ms[1]:.4f} tail {ms[2]:.4f} ms; network {e0.elapsed_time(e1) / 20:.4f} ms per 12 planes; crc {float(y.sum()):.6f}", flush=True)
    sys.exit(0)
var, vals = sys.argv[1], sys.argv[2:]
for r in range(3):
    for v in vals:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, **{var: v}), capture_output=True, text=True, timeout=600)
        for l in o.stdout.splitlines():
            if l.startswith("RES"): print(f"round {r} {var}={v}: {l[4:]}", flush=True)
        if o.returncode != 0: print(o.stderr[-800:])
