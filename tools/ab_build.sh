#!/bin/bash
# A/B of the current build against another build of the repo kept under tools/_ab_old/ (git-ignored; made with
#   git worktree add /tmp/old <commit> && make -j8 -C /tmp/old/video-moment-localization_amd/csrc, then bench.py models.py include oracle
#   video-moment-localization_amd BASELINE.json profiles/pmc_moment.json copied into tools/_ab_old/)
# on ONE box, alternating, three runs each.   usage: tools/ab_build.sh [bench args...]
R=$(pwd)
for i in 1 2 3; do
  for d in tools/_ab_old .; do
    (cd $R/$d && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-modes "$@" 2>/dev/null) | python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('[%s]' % '$d', round(d['ms_per_step'],3), round(d['ms_fwd_bwd'],3))"
  done
done
