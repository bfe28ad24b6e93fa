#!/bin/bash
# Run ON THE GPU BOX: the headline-path tests that compare the fp16 production path with the oracle record what they measured
# (PSNR, max LSB) in gpurun_out/parity_measured.json (tests/helpers.py record_measured); copy it to profiles/<tag>_parity_measured.json.
# usage: bash tools/parity_measured.sh r03
set -u
TAG=${1:-r03}
cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_measured.json
python3 -m pytest tests/test_gpu_headline.py -q -m gpu -k "vs_oracle or literal_tolerance" 2>&1 | tail -5
cp gpurun_out/parity_measured.json gpurun_out/${TAG}_parity_measured.json
cat gpurun_out/${TAG}_parity_measured.json
