#!/bin/bash
# Run ON THE GPU BOX: FSRCNN x2 720p (4 frames per step), both matrix-core modes, with and without an environment switch of the library
# (one process per run, interleaved).   usage: bash tools/fs_env_ab.sh "SS4K_MH_NO_TALL=1 SS4K_TAIL_NO_TALL=1" [rounds=3]
SW=$1; R=${2:-3}
one() {  # $1 = label, $2 = environment, $3 = workload
  env $2 python3 bench.py --workload $3 --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-11s %7.1f frames/s   stages ms: %s' % ('$1', '$3', d['value'], '  '.join('%s %.4f' % (k.split(' ')[0], v['ms_per_step']) for k, v in d['roofline']['stages'].items())))"
}
BASE=${3:-SS4K_AB=0}   # third argument: environment of BOTH runs (e.g. SS4K_LIB=... for switches of the dev library)
for r in $(seq $R); do for wl in fsrcnn_f16 fsrcnn; do one with "$BASE $SW" $wl; one without "$BASE" $wl; done; done
