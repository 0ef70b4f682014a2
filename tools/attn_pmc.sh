#!/bin/bash
# PMC passes over the attention micro-benchmark (run on the GPU box from the repo root): bash tools/attn_pmc.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/attn_pmc}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/tools/attn_bench.py --iters 3 > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $ROOT && python3 profiles/pmc_summary.py $OUT | grep -A1 "content_attn" 
