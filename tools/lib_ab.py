"""Dev tool: the same RRDBNet x2 720p job under DIFFERENT BUILDS of the library, on one box.  One child process per (round,
library) - a process binds one libss4k_hip*.so - interleaved so that box drift hits all builds alike.
usage: python tools/lib_ab.py <frames> <rounds> <name=path/to/lib.so> [<name=path> ...]     (run on the GPU box)
       python tools/lib_ab.py --child <frames>                                            (internal: prints fps)"""
import os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    import sharkshark4k_amd  # noqa
    from sharkshark4k_amd import _capi, weights as W
    nf = int(sys.argv[2])
    ctx = _capi.Context(0)
    flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    up = _capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0)
    frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (nf, 720, 1280, 3), dtype=np.uint8)).cuda()
    out = torch.empty((nf, 1440, 2560, 3), dtype=torch.uint8, device="cuda")
    for _ in range(8):
        up(frames, out)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(25):
            up(frames, out)
        torch.cuda.synchronize()
        best = max(best, 25 * nf / (time.perf_counter() - t0))
    print(f"FPS {best:.2f} CRC {int(out.to(torch.int64).sum())}")
    sys.exit(0)

nf, rounds = int(sys.argv[1]), int(sys.argv[2])
libs = [a.split("=", 1) for a in sys.argv[3:]]
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for n, path in libs:
        env = dict(os.environ, SS4K_LIB=os.path.abspath(path))
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(nf)], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in o.stdout.splitlines() if l.startswith("FPS")]
        if o.returncode != 0 or not line:
            print(f"round {r} {n}: FAILED rc {o.returncode}\n{o.stderr[-1500:]}", flush=True); continue
        res[n].append(float(line[0].split()[1]))
        print(f"round {r} {n:>10s}: {line[0]}", flush=True)
for n, v in res.items():
    if v:
        print(f"{n:>10s}: {nf}-frame jobs, best-of-3 per process over {len(v)} processes: min {min(v):.2f}  median {sorted(v)[len(v) // 2]:.2f}  max {max(v):.2f} fps")
