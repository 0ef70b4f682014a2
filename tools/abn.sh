#!/bin/bash
# A/B/C... of several environment settings on ONE box, alternating, N rounds:  tools/abn.sh <rounds> "<env1>" "<env2>" ...
N=$1; shift
for i in $(seq $N); do
  for cfg in "$@"; do
    env $cfg timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('[%s]' % '$cfg', round(d['ms_per_step'],3), round(d['ms_fwd_bwd'],3))"
  done
done
