"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel into a JSON / table (tracked script that
produces the profiles/*_sq_counters.json files).
usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [out.json]

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles
(MI355X_MICROARCH.md, cycle-constants table), so the matrix-pipe share of a wave's life is
MFMA_BUSY / (4 * WAVE_CYCLES); x waves per SIMD = share of SIMD time."""
import collections, csv, glob, json, os, sys

src = sys.argv[1]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(src) else [src]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for f in files:
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); launches[k] += 1
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    e = {"launches": launches[k], **{c: x for c, x in v.items()}}
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY"),
                        ("wait_inst_lds", "SQ_WAIT_INST_LDS")):
            if c in v: e[name + "_share"] = v[c] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v: e["mfma_busy_per_wave_share"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)
    if v.get("SQ_LDS_IDX_ACTIVE"): e["lds_conflict_share_of_lds_cycles"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
    if "TCC_HIT_sum" in v: e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(1.0, v["TCC_HIT_sum"] + v.get("TCC_MISS_sum", 0.0))
    out[k] = e
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, e in list(out.items())[:12]:
    print(k[:72].ljust(72), " ".join(f"{n}={x:.3f}" for n, x in e.items() if n.endswith("share") or n.endswith("rate") or n.endswith("cycles")), f"n={e['launches']}")
