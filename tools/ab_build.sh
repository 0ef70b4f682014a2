#!/bin/bash
# A/B of the current build against another build of the repo kept under tools/_ab_old/ (git-ignored; tools/ab_snapshot.sh keeps the
# current build there before a change) on ONE box, alternating, three runs each.   usage: tools/ab_build.sh [bench args...]
R=$(pwd)
for i in 1 2 3; do
  for d in tools/_ab_old .; do
    (cd $R/$d && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes "$@" 2>/dev/null) | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); g=d['roofline']['moment_gemms']; print('[%s]' % '$d', round(d['ms_per_step'],3), round(d['ms_fwd_bwd'],3), 'fwd/dx/dw TF', round(g['moment_fwd']['achieved'],1), round(g['moment_dx']['achieved'],1), round(g['moment_dw']['achieved'],1), 'alone', round(g['moment_fwd']['alone_tflops'],1), round(g['moment_dx']['alone_tflops'],1), round(g['moment_dw']['alone_tflops'],1))"
  done
done
