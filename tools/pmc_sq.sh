#!/bin/bash
# SQ counter passes for a small script (GPU box): bash tools/pmc_sq.sh <script.py> [kernel-substring]
export TMPDIR=/tmp
out=gpurun_out/pmcsq; rm -rf $out; mkdir -p $out
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/a -o a -- python3 $1 > /dev/null 2>&1
timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $out/b -o b -- python3 $1 > /dev/null 2>&1
python tools/pmc_summary.py $out "$2"
rm -rf $out
