#!/bin/bash
# like ab.sh but prints the attention launch times
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['roofline']['attention_core']; print('[%s]' % '$cfg', round(d['ms_per_step'],3), 'attn fwd', round(a['attn_fwd']['avg_launch_ms'],4), 'bwd', round(a['attn_bwd']['avg_launch_ms'],4), 'dx', round(d['roofline']['moment_gemms']['moment_dx']['avg_launch_ms'],4))"
  done
done
