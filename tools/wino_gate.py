"""Round-5 gate of VERDICT item 4 (run ON THE GPU BOX): the Winograd skeletons of tools/micro/wino_skeleton.hip beside the REAL direct kernels on
the same layer shapes, each run for seconds on random data so that the chip sits at its power cap, with sclk / W sampled beside it.
Direct kernels through the dev library's ss4k_bench_conv (one layer alone on the chip, 2 frames of 360x640 = what one launch chain of the
headline job launches): conv5 192 -> 64 on conv_w16.hip<RL>, trunk 64 -> 64 on conv_w16.hip.
usage: python3 tools/wino_gate.py [seconds per arm]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SS4K_LIB", os.path.join(ROOT, "sharkshark-4k_amd", "libss4k_hip_dev.so"))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0

samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = [l.split("(")[-1].split("Mhz")[0] for l in o.splitlines() if "sclk" in l]
            pw = [l.split(":")[-1].strip() for l in o.splitlines() if "Power (W)" in l]
            samples.append((time.time(), sclk[0] if sclk else "?", pw[0] if pw else "?"))
        except Exception:
            pass
        time.sleep(0.4)
th = threading.Thread(target=sampler, daemon=True); th.start()
def window(t0, t1):
    s = [x for x in samples if t0 + 1.0 <= x[0] <= t1]
    return (f"sclk {min(int(x[1]) for x in s)}-{max(int(x[1]) for x in s)} MHz, {min(float(x[2]) for x in s):.0f}-{max(float(x[2]) for x in s):.0f} W"
            if s and all(x[1].isdigit() for x in s) else "no samples")

import sharkshark4k_amd  # noqa: F401,E402
from sharkshark4k_amd import _capi  # noqa: E402
ctx = _capi.Context(0)
H, W, n = 360, 640, 2
for name, c0, c1, co, fl in (("direct conv_w16<RL>  192 -> 64 (conv5)", 64, 128, 64, 2048), ("direct conv_w16       64 -> 64 (trunk)", 64, 0, 64, 0)):
    us0 = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, flags=fl, iters=50)
    iters = max(100, int(secs * 1e6 / us0))
    t0 = time.time(); us = ctx.bench_conv(_capi.F16, c0, c1, co, n, H, W, flags=fl, iters=iters); t1 = time.time()
    gf = 2 * 9 * (c0 + c1) * co * n * H * W / 1e9
    print(f"{name:78s} {gf / us * 1e3:7.0f} direct-equivalent TFLOP/s  ({us:.1f} us per 2-frame launch; {window(t0, t1)})", flush=True)
exe = os.path.join(ROOT, "tools", "micro", "wino_skeleton")
t0 = time.time()
p = subprocess.Popen([exe, str(secs)], stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stdout:
    marks.append((time.time(), line.rstrip()))
p.wait()
prev = t0
for t, line in marks:
    print(f"{line}  [{window(prev, t)}]", flush=True)
    prev = t
stop = True
