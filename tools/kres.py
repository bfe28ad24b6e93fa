"""Dev tool: registers / scratch / LDS of every gfx950 kernel in csrc/ as hipcc reports them (-Rpass-analysis=kernel-resource-usage,
device-only compile, the product flags of build.py).  No GPU needed.
usage: python tools/kres.py [--dev] [--all] [file.hip ...]      (default: every .hip of build.py's SOURCES; --all: also kernels without scratch)"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import build as B

dev = "--dev" in sys.argv
show_all = "--all" in sys.argv
files = [a for a in sys.argv[1:] if not a.startswith("--")] or [s for s in B.SOURCES + (B.DEV_SOURCES if dev else []) if s.endswith(".hip")]
keys = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"]
print(f"{'kernel':92s} {'vgpr':>4s} {'agpr':>4s} {'scratch':>7s} {'sspill':>6s} {'vspill':>6s} {'lds':>6s} {'occ':>3s}")
for f in files:
    src = os.path.join(B.CSRC, f)
    flags = [x for x in B.FLAGS if x != "-fPIC"] + (["-DSS4K_DEV"] if dev else []) + B.EXTRA_FLAGS.get(f, [])
    r = subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stderr); sys.exit(1)
    demangle = {}
    cur = None; rows = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}; continue
        for k in keys:
            m = re.search(r"remark:\s+" + re.escape(k) + r":\s+(\d+)", line)
            if m and cur:
                rows[cur][k] = int(m.group(1))
    names = list(rows)
    if names:
        d = subprocess.run(["c++filt", *names], capture_output=True, text=True).stdout.splitlines()
        demangle = dict(zip(names, d))
    for n, v in rows.items():
        if not show_all and not v.get(keys[2]) and not v.get(keys[3]) and not v.get(keys[4]):
            continue
        nm = re.sub(r"\(.*$", "", demangle.get(n, n)).replace("void ", "").replace("ss4k::", "")
        print(f"{f + ': ' + nm:92.92s} {v.get(keys[0], 0):4d} {v.get(keys[1], 0):4d} {v.get(keys[2], 0):7d} {v.get(keys[3], 0):6d} {v.get(keys[4], 0):6d} "
              f"{v.get(keys[5], 0):6d} {v.get(keys[6], 0):3d}")
