#!/bin/bash
# Run ON THE GPU BOX: every GPU parity test that compares a production path with the oracle (or one mode with another) records
# what it measured (PSNR, max LSB, share of differing bytes) in gpurun_out/parity_measured.json (tests/helpers.py
# record_measured).  This script runs the WHOLE gpu marker set from an empty accumulator, so the tracked copy always holds
# every entry (a later single-test run appends to / refreshes gpurun_out/parity_measured.json but never the tagged copy).
# usage: bash tools/parity_measured.sh r04   ->  gpurun_out/r04_parity_measured.json  (copy it to profiles/)
set -u
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_measured.json
python3 -m pytest tests -q -m gpu -x 2>&1 | tail -5
cp gpurun_out/parity_measured.json gpurun_out/${TAG}_parity_measured.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_parity_measured.json"))
print(len(d), "entries:", ", ".join(sorted(d)))
PY
