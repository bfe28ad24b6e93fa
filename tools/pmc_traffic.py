"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_conv3x3_pmc_traffic.json.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [frames_per_launch]
Counts the dispatches of both 3x3 conv kernels (conv3x3_kernel = LDS weights, conv3x3_rs_kernel = register-stationary
weights) - the same launches bench.py's roofline leg times; FETCH_SIZE is doubled per MI355X_MICROARCH.md (HBM section)."""
import csv, glob, json, sys
from collections import defaultdict


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
            FAMILY[r["Dispatch_Id"]] = "conv3x3_rs_kernel" if "conv3x3_rs_kernel" in r["Kernel_Name"] else "conv3x3_kernel"
    return acc


FAMILY = {}


fetch_d = per_dispatch(sys.argv[1], "FETCH_SIZE")
fam_f = dict(FAMILY); FAMILY.clear()
write_d = per_dispatch(sys.argv[2], "WRITE_SIZE")
fam_w = dict(FAMILY)
fetch, write = list(fetch_d.values()), list(write_d.values())
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
fa, wa = sum(fetch) / len(fetch), sum(write) / len(write)
by_family = {}
for fam in sorted(set(fam_f.values())):
    ff = [v for k, v in fetch_d.items() if fam_f[k] == fam]; ww = [v for k, v in write_d.items() if fam_w.get(k) == fam]
    if ff and ww:
        by_family[fam] = {"launches": len(ff), "traffic_bytes_per_launch": (2 * sum(ff) / len(ff) + sum(ww) / len(ww)) * 1024}
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline --no-roofline --no-also (RRDBNet x2 720p fp16, %d frames per launch), launches of the two 3x3 conv kernels" % n,
    "launches_counted": len(fetch), "frames_per_launch": n,
    "fetch_size_kb_avg_raw": fa, "write_size_kb_avg": wa,
    "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; KB x1024",
    "traffic_bytes_per_launch": (2 * fa + wa) * 1024,
    "by_kernel": by_family,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
