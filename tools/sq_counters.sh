#!/bin/bash
# SQ counter passes for the step's kernels (separate --pmc runs, no trace domains): tools/sq_counters.sh <tag>
set -e -o pipefail
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py ${PNPP_SQ_EXTRA:-} --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-graph --no-bf16-variant --no-mfma-variant"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS -d $OUT/p1 -o p1 --output-format csv -- python3 $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/p2 -o p2 --output-format csv -- python3 $B > $OUT/p2.log 2>&1
python3 tools/pmc_table.py $OUT/p1 "${PNPP_SQ_FILTER-gemm_}" > $OUT/sq_p1.txt
python3 tools/pmc_table.py $OUT/p2 "${PNPP_SQ_FILTER-gemm_}" > $OUT/sq_p2.txt
rm -rf $OUT/p1 $OUT/p2
cat $OUT/sq_p1.txt $OUT/sq_p2.txt
