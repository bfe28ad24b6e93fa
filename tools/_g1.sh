set -e
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
