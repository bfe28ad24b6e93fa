#!/bin/bash
# Run ON THE GPU BOX: the round-3 additions (FSRCNN fp16 mode, fused BSVD pairs): per-kernel stats of the two workloads, SQ counters
# of their kernels, the fused-vs-two-launches A/B and the FSRCNN mode table.  usage: bash tools/collect_profiles_r03c.sh
set -u
TAG=r03c
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
for wl in fsrcnn_f16 pipeline; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --workload $wl --steps 8 --warmup 6 --no-cpu-baseline --no-also --no-roofline > $O/bench_$wl.log 2>&1
  cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/${TAG}_${wl}_kernel_stats.csv
  rm -rf $O/st
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sqf -- python3 bench.py --workload fsrcnn_f16 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sqf $O/${TAG}_fsrcnn_f16_sq_counters.json > /dev/null
rm -rf $O/sqf
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sqb -- python3 tools/bsvd_trace.py 4 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sqb $O/${TAG}_bsvd_sq_counters.json > /dev/null
rm -rf $O/sqb
python3 tools/bsvd_ab.py 4 3 > $O/${TAG}_bsvd_pair_ab.txt 2>&1
python3 tools/bsvd_ab.py 1 3 >> $O/${TAG}_bsvd_pair_ab.txt 2>&1
python3 tools/fs_modes.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_fsrcnn_modes.txt
ls -la $O
