"""Per-launch table of the last N kernel launches of a rocprofv3 --kernel-trace run (name, us, grid, VGPRs, scratch).
usage: python tools/trace_table.py <dir or *_kernel_trace.csv> [N] [name filter]"""
import csv, glob, os, sys
src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0] if os.path.isdir(src) else src
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
tot = 0.0
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    nm = r["Kernel_Name"].replace("void ss4k::", "").replace("ss4k::", "")[:64]
    print(f"{d:9.1f} us  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<3} vgpr {r['VGPR_Count']:>3}+{r['Accum_VGPR_Count']:<3} scratch {r['Scratch_Size']:>4}  {nm}")
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"sum of kernels {tot:.1f} us, span {span:.1f} us over {len(last)} launches")
