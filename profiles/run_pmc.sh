#!/bin/bash
# PMC passes over bench.py (separate rocprofv3 runs, counters only + kernel trace; see MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage (on the GPU box, from the repo root):  [BENCH_ARGS="--gemm f32e"] bash profiles/run_pmc.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes $BENCH_ARGS > $ROOT/$OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $ROOT/$OUT/*/*/ | head -40
