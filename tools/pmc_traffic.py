"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_conv3x3_pmc_traffic.json.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [frames_per_launch]
Counts conv3x3_kernel dispatches only; FETCH_SIZE is doubled per MI355X_MICROARCH.md (HBM section)."""
import csv, glob, json, sys
from collections import defaultdict


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "conv3x3_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return list(acc.values())


fetch = per_dispatch(sys.argv[1], "FETCH_SIZE")
write = per_dispatch(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[4]) if len(sys.argv) > 4 else 4
fa, wa = sum(fetch) / len(fetch), sum(write) / len(write)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline --no-roofline --no-also (RRDBNet x2 720p fp16, %d frames per launch), conv3x3_kernel launches only" % n,
    "launches_counted": len(fetch), "frames_per_launch": n,
    "fetch_size_kb_avg_raw": fa, "write_size_kb_avg": wa,
    "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; KB x1024",
    "traffic_bytes_per_launch": (2 * fa + wa) * 1024,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
