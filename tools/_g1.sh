set -e
export SS4K_LIB=$PWD/sharkshark-4k_amd/libss4k_hip_dev.so
timeout -k 10 600 python -m pytest tests/test_gpu_dense_pair.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 400 python tools/env_ab.py "SS4K_DENSE_MASK=0;SS4K_DENSE_MASK=1;SS4K_DENSE_MASK=2;SS4K_DENSE_MASK=3" 4 3 2>&1 | tee gpurun_out/r04_dense_mask_n4.txt
timeout -k 10 400 python tools/env_ab.py "SS4K_DENSE_MASK=0;SS4K_DENSE_MASK=1;SS4K_DENSE_MASK=2;SS4K_DENSE_MASK=3" 2 3 2>&1 | tee gpurun_out/r04_dense_mask_n2.txt
timeout -k 10 400 python tools/env_ab.py "SS4K_DENSE_MASK=0;SS4K_DENSE_MASK=1;SS4K_DENSE_MASK=2;SS4K_DENSE_MASK=3" 1 3 2>&1 | tee gpurun_out/r04_dense_mask_n1.txt
