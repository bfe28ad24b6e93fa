for i in 1 2; do for q in 8 4; do GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python bench.py --no-by-kernel > gpurun_out/hwq_${q}_$i.json 2> gpurun_out/hwq_${q}_$i.err || exit 1; python3 - <<PY
import json
d=json.load(open("gpurun_out/hwq_${q}_$i.json")); a=d["also"]
print("queues $q run $i: headline %.1f pipeline %.1f srvgg %.1f n1 %.1f n1_one_set %.1f worker %.1f fsrcnn %.0f fsrcnn_f16 %.0f x4 %.2f" % (d["value"], a["pipeline"]["fps"], a["srvgg"]["fps"], a["rrdbnet_n1"]["fps"], a["rrdbnet_n1_one_set"]["fps"], a["rrdbnet_n1_worker"]["fps"], a["fsrcnn"]["fps"], a["fsrcnn_f16"]["fps"], a["rrdbnet_x4"]["fps"]))
PY
grep -c "fails the pair test" gpurun_out/hwq_${q}_$i.err; done; done
