#!/bin/bash
# A/B of two environment settings on ONE box (boxes of the pool differ by several per cent): alternating runs, three each.
# usage: tools/ab.sh "<env A>" "<env B>" [bench args...]
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('[%s]' % '$cfg', round(d['ms_per_step'],3), round(d['ms_fwd_bwd'],3), 'dx', round(d['roofline']['moment_gemms']['moment_dx']['avg_launch_ms'],4))"
  done
done
