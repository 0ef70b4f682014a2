#!/bin/bash
# step time of each contraction mode on one box
for m in f32 f32e bf16x3 bf16; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes --gemm $m "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); g=d['roofline']['moment_gemms']; print('[$m]', round(d['ms_per_step'],3), round(d['ms_fwd_bwd'],3), 'fwd/dx/dw', round(g['moment_fwd']['avg_launch_ms'],4), round(g['moment_dx']['avg_launch_ms'],4), round(g['moment_dw']['avg_launch_ms'],4), 'loss', d['config']['final_loss'])"
done
