"""Dev tool: print a rocprofv3 --stats kernel_stats.csv (name, calls, average us, share).  usage: python tools/kstats.py <dir|csv> [N]"""
import csv, glob, os, sys
src = sys.argv[1]
f = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0] if os.path.isdir(src) else src
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    nm = r["Name"].replace("void ss4k::", "").replace("ss4k::", "").split("(")[0][:70]
    print(f"{float(r['AverageNs']) / 1e3:10.1f} us x {int(r['Calls']):5d}  {float(r['Percentage']):6.2f} %  {nm}")
