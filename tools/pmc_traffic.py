"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<round>_conv3x3_pmc_traffic.json.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> <frames_per_launch> <steps_run> <frames_per_step>
Counts the dispatches of the 3x3 conv kernels (conv3x3_kernel = LDS weights, conv3x3_rs_kernel = register-stationary weights,
conv3x3_dense2_kernel = fused dense-block layer pairs, conv3x3_wide_kernel / conv3x3_w16_kernel = one 64-cout layer on the fused kernel's machinery, 32x32x16 / 16x16x32 MFMA) - the same launches bench.py's roofline leg times; FETCH_SIZE is doubled per
MI355X_MICROARCH.md (HBM section).  The figure bench.py uses is bytes per STEP (all conv launches of a step together): the
kernels differ too much for a per-launch average to mean anything."""
import csv, glob, json, sys
from collections import defaultdict

# most specific name first; a conv3x3 kernel none of them matches is counted under its own name (a new kernel must not crash the collection)
FAMILIES = ("conv3x3_rs_kernel", "conv3x3_dense2_kernel", "conv3x3_w16n_kernel", "conv3x3_w16_kernel", "conv3x3_wide_kernel", "conv3x3_kernel")


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, fam = defaultdict(float), {}
    for r in csv.DictReader(open(f)):
        if "conv3x3" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
            fam[r["Dispatch_Id"]] = next((k for k in FAMILIES if k in r["Kernel_Name"]), r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1])
    return acc, fam


fetch_d, fam_f = per_dispatch(sys.argv[1], "FETCH_SIZE")
write_d, fam_w = per_dispatch(sys.argv[2], "WRITE_SIZE")
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
fps = int(sys.argv[6]) if len(sys.argv) > 6 else 4
fetch_kb, write_kb = sum(fetch_d.values()), sum(write_d.values())
total = (2 * fetch_kb + write_kb) * 1024
by_family = {}
for fam in list(FAMILIES) + sorted(set(fam_f.values()) - set(FAMILIES)):
    ff = [v for k, v in fetch_d.items() if fam_f[k] == fam]; ww = [v for k, v in write_d.items() if fam_w.get(k) == fam]
    if ff and ww:
        by_family[fam] = {"launches_per_step": len(ff) / steps, "traffic_bytes_per_launch": (2 * sum(ff) / len(ff) + sum(ww) / len(ww)) * 1024,
                          "traffic_bytes_per_step": (2 * sum(ff) + sum(ww)) * 1024 / steps}
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline --no-roofline --no-also --no-by-kernel (RRDBNet x2 720p fp16, %d frames per step, %d per launch: SS4K_LANES=2; "
              "%d forwards in all with the settle calls of bench.py), every launch of the 3x3 conv kernels" % (fps, n, steps),
    "launches_counted": len(fetch_d), "steps": steps, "frames_per_step": fps, "frames_per_launch": n,
    "correction": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is; KB x1024",
    "traffic_bytes_per_step": total / steps,
    "traffic_bytes_per_launch": total / len(fetch_d),
    "by_kernel": by_family,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
