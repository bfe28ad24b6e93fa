#!/bin/bash
# Run ON THE GPU BOX (gpurun): collects the round's rocprofv3 evidence into gpurun_out/<tag>/ ; the summaries are then
# copied into profiles/ (tracked).  usage: bash tools/collect_profiles.sh r02a [sections, e.g. "4,5" - default all: 1 2 3 4 5 6]
# PMC counters go in their own runs with --kernel-trace only (never with sys/hip traces: gpurun refuses that).
set -u
TAG=${1:-r02}
ONLY=${2:-all}
want() { [ "$ONLY" = all ] || [[ ",$ONLY," == *",$1,"* ]]; }
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also --no-by-kernel --no-live-traffic"
if want 1; then
# 1. headline: per-kernel stats + the bench line printed under the profiler + how many conv launches are in flight
#    (frame lanes: two concurrent launch chains; the choice is measured by the library over its first calls)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- $B --steps 8 --warmup 6 > $O/bench_under_rocprof.log 2>&1
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/${TAG}_rrdbnet_x2_720p_batch4_kernel_stats.csv
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/${TAG}_bench_line_under_rocprof.json
python3 tools/trace_overlap.py $O/st 0.5 > $O/${TAG}_rrdbnet_lanes_overlap.txt
rm -rf $O/st
# 1b. the same job as ONE launch chain (SS4K_LANES=1, read when a model is built): every launch alone on the chip, so that a kernel's
#     average duration is its own (with two chains it includes the time it shares the chip with the other chain's launch)
export SS4K_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st1 -- $B --steps 8 --warmup 6 > $O/bench_one_chain_under_rocprof.log 2>&1
cp $(find $O/st1 -name "*kernel_stats.csv" | head -1) $O/${TAG}_rrdbnet_x2_720p_batch4_one_chain_kernel_stats.csv
grep '^{' $O/bench_one_chain_under_rocprof.log | tail -1 > $O/${TAG}_bench_line_one_chain_under_rocprof.json
rm -rf $O/st1
unset SS4K_LANES
fi
if want 2; then
# the counter passes serialise kernels: two chains are forced (SS4K_LANES=2) so that every launch carries 2 frames
export SS4K_LANES=2
# 2. headline: fabric traffic of the conv launches (separate passes)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B --steps 2 --warmup 1 --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B --steps 2 --warmup 1 --no-roofline > /dev/null 2>&1
# (steps run = 7 settle calls of the lane tuning + 1 warm-up + 2 timed: bench.py)
python3 tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_conv3x3_pmc_traffic.json 2 10 4 > /dev/null
# the name bench.py reads (bench.py PMC_TRAFFIC_FILE): both files are copied into profiles/
cp $O/${TAG}_conv3x3_pmc_traffic.json $O/conv3x3_pmc_traffic_current.json
rm -rf $O/fetch $O/write
fi
if want 3; then
# 3. headline: SQ counters (matrix-pipe share, LDS conflicts, wait shares) and L2 hit rate, per kernel
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sq -- $B --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sq $O/${TAG}_rrdbnet_sq_counters.json > /dev/null
rm -rf $O/sq
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/tcc -- $B --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/tcc $O/${TAG}_rrdbnet_tcc_counters.json > /dev/null
rm -rf $O/tcc
# 3b. where the fabric requests go: rocprofv3 on gfx950 lists no Infinity-Cache (MALL) hit / miss counter (rocprofv3 -L: nothing
#     named mall / l3); the nearest split the TCC exposes is "requests destined for DRAM (MC)" against all EA requests
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum --kernel-trace --output-format csv -d $O/ea -- $B --steps 1 --warmup 1 --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/ea $O/${TAG}_rrdbnet_ea_dram_counters.json > /dev/null
rm -rf $O/ea
unset SS4K_LANES
# 3c. working set against the Infinity Cache, WITH frame lanes: 4 frames at once (default) against two passes of 2 frames
#     (dev library switch SS4K_SUBBATCH=2; each pass still runs as two launch chains of one frame)
SS4K_LIB=$PWD/sharkshark-4k_amd/libss4k_hip_dev.so python3 tools/env_ab.py "SS4K_LANES=2,SS4K_SUBBATCH=0;SS4K_LANES=2,SS4K_SUBBATCH=2" 4 3 > $O/${TAG}_subbatch_ab_with_lanes.txt 2>&1
fi
if want 4; then
# 4. the other workloads: per-kernel stats
for wl in fsrcnn fsrcnn_f16 pipeline srvgg rrdbnet_x4; do
  extra=""; [ $wl = rrdbnet_x4 ] && extra="--batch 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --workload $wl --steps 8 --warmup 6 --no-cpu-baseline --no-also --no-roofline $extra > $O/bench_$wl.log 2>&1
  cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/${TAG}_${wl}_kernel_stats.csv
  rm -rf $O/st
done
fi
if want 5; then
# 5. FSRCNN: SQ counters of its kernels
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sqf -- python3 bench.py --workload fsrcnn --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sqf $O/${TAG}_fsrcnn_sq_counters.json > /dev/null
rm -rf $O/sqf
# 5b. ... and of the fp16 mode's kernels (k_fs_head_m<false>, k_fs_maps4_h, k_fs_tail_r<..., false, true>)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sqh -- python3 bench.py --workload fsrcnn_f16 --steps 1 --warmup 1 --no-cpu-baseline --no-also --no-roofline > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sqh $O/${TAG}_fsrcnn_f16_sq_counters.json > /dev/null
rm -rf $O/sqh
fi
if want 6; then
# 6. the fused dense-block pairs on v_mfma_f32_16x16x32_f16 (conv_d16.hip, dev library) against the default 32x32x16 build, on this binary:
#    interleaved A/B at 4 / 2 / 1 frames, its phase stamps, and its own duration per launch from a one-chain bench run
DEV=$PWD/sharkshark-4k_amd/libss4k_hip_dev.so
{ for b in 4 2 1; do SS4K_LIB=$DEV python3 tools/env_ab.py "SS4K_D16=0;SS4K_D16=1" $b 3 2>&1 | grep -v "^HipUpscalerService"; done
  SS4K_LIB=$DEV SS4K_D16=1 SS4K_D16_STAMP=1 python3 tools/env_ab.py "SS4K_D16=1" 4 1 2>&1 | grep "d16 K1" | head -4
  for d in 0 1; do SS4K_LIB=$DEV SS4K_D16=$d python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-live-traffic 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print('SS4K_D16=$d: headline %.1f frames/s, conv frac %.3f; one chain, per kernel build:' % (d['value'], d['roofline']['frac']))
for k in bk['kernels'][:4]: print('   %-60s %6.1f us per 4-frame launch  %.3f of peak  %.1f %% of kernel time' % (k['kernel'][:60], k['avg_launch_us'], k['frac'], 100*k['share_of_kernel_time']))
"; done; } > $O/${TAG}_d16_gate.txt 2>&1
fi
ls -la $O
