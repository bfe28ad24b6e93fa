#!/bin/bash
# Run ON THE GPU BOX: runtime-setting A/B on the headline job, one process per value, interleaved.
# usage: bash tools/kernarg_ab.sh [VAR=HIP_FORCE_DEV_KERNARG] [values="0 1"] [extra bench args]
VAR=${1:-HIP_FORCE_DEV_KERNARG}; VALS=${2:-"0 1"}; shift; shift
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for v in $VALS; do
    echo -n "round $r $VAR=$v: "
    env $VAR=$v python3 bench.py --no-cpu-baseline --no-by-kernel --steps 40 --warmup 10 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d.get('also',{}); print(round(d['value'],2), round(d['roofline']['frac'],4), {k:round(v['fps'],1) for k,v in a.items() if 'fps' in v})"
  done
done
