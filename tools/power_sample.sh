#!/bin/bash
# Run ON THE GPU BOX: sample sclk and socket power with rocm-smi while a workload runs (is the part at its power cap? at what clock?).
# usage: bash tools/power_sample.sh headline            -> profiles/earlier/r04/r04_headline_power_clock.txt style output on stdout
#        bash tools/power_sample.sh srvgg <model flags> -> the SRVGG x4 720p network in a loop (tools/power_loop.py), e.g. 0 and 32768 (NO_W16)
set -u
cd "$GRAFT_REPO_ROOT"
if [ "${1:-headline}" = headline ]; then
  python3 bench.py --steps 1000 --warmup 20 --no-also --no-cpu-baseline > /tmp/power_bench.json 2> /dev/null &
else
  python3 tools/power_loop.py "${2:-0}" 8 > /tmp/power_bench.json 2>&1 &
fi
BP=$!
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>&1 | grep -E 'sclk|Power \(W\)' | sed 's/.*(\([0-9]*Mhz\)).*/\1/; s/.*(W): //' | tr '\n' ' ')"
  sleep 0.5
done
wait $BP
tail -c 400 /tmp/power_bench.json
