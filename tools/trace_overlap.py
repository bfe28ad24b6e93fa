"""Dev tool: concurrency of the conv launches in a rocprofv3 --kernel-trace run (frame lanes).  Per queue: launches,
mean duration by kernel; overall: busy time with >= 1 / >= 2 conv kernels in flight over the last `frac` of the trace.
usage: python tools/trace_overlap.py <dir or *_kernel_trace.csv> [frac=0.3]"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0] if os.path.isdir(src) else src
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = [r for r in csv.DictReader(open(f)) if "conv3x3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
byq = defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
for q, rs in byq.items():
    dur = defaultdict(list)
    for r in rs:
        nm = r["Kernel_Name"].replace("void ss4k::", "").split("(")[0][:60]
        dur[(nm, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rs, rs[1:])]
    print(f"queue {q}: {len(rs)} launches, mean gap to the next launch {sum(gaps) / max(1, len(gaps)):.2f} us")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print(f"    {sum(v) / len(v):8.1f} us x {len(v):5d}  grid {k[1]:>6}  {k[0]}")
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
lvl, last, acc = 0, t0, defaultdict(int)
for t, d in ev:
    acc[lvl] += t - last; last = t; lvl += d
tot = t1 - t0
print(f"span {tot / 1e6:.2f} ms: " + "  ".join(f"{k} in flight {100.0 * v / tot:.1f} %" for k, v in sorted(acc.items())))
