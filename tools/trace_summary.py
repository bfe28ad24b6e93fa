import csv, glob, sys, numpy as np
d0 = sys.argv[1]
f=glob.glob(d0+'/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'conv3x3' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-351:]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in last]
print("conv_first", d[0], "vgpr", last[1]['VGPR_Count'], last[1]['Accum_VGPR_Count'], "scratch", last[1]['Scratch_Size'], "lds", last[1]['LDS_Block_Size'], "grid", last[1]['Grid_Size_X'])
body=np.array(d[1:1+345]).reshape(69,5)
print("per-conv mean us:", body.mean(0).round(1), "sum", body.mean(0).sum().round(1))
gf=np.array([8.5,12.7,17.0,21.2,51.0])
print("TFLOP/s per conv:", (gf/body.mean(0)*1e3).round(0))
print("tail:", d[346:])
print("total conv ms", sum(d)/1e3)
