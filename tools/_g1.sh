set -e
timeout -k 10 600 python -m pytest tests/test_gpu_dense_pair.py -x -q -m gpu 2>&1 | tail -3
python tools/lib_ab.py 4 3 v2b=build_variants/libss4k_hip_dev_v2b.so v2c=sharkshark-4k_amd/libss4k_hip_dev.so
python tools/lib_ab.py 1 2 v2b=build_variants/libss4k_hip_dev_v2b.so v2c=sharkshark-4k_amd/libss4k_hip_dev.so
