#!/bin/bash
# same-box A/B of one environment setting on the step of other BASELINE configurations (two rounds each, interleaved):
#   tools/ab/ab_env_cfg.sh SEGFAC_GEMM8_LINEAR=0 "cfg3 --batch 32" "cfg5 --batch 8" "cfg5 --batch 8 --fp8"
cd $GRAFT_REPO_ROOT
SET=$1; shift
for c in "$@"; do for v in 0 1 0 1; do
  if [ $v = 1 ]; then export $SET; else unset ${SET%%=*}; fi
  python3 bench.py --config $c --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', '$SET' if $v else 'default', d['value'], d['ms_per_step'])"
done; done
