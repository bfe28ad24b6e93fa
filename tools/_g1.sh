set -e
python tools/lib_ab.py 4 3 rb8=build_variants/libss4k_hip_rb8.so rb4=sharkshark-4k_amd/libss4k_hip.so
