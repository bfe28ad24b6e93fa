# SQ counters of the BSVD kernels (fused layer pairs included); usage: bash tools/bsvd_pmc.sh
export TMPDIR=/tmp
O=gpurun_out/bsvd_pmc
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq -- python3 tools/bsvd_trace.py 4 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sq $O/bsvd_sq.json
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- python3 tools/bsvd_trace.py 4 > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sq2 $O/bsvd_sq2.json
