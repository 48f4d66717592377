#!/bin/bash
# same-box on / off of one environment switch on the step of other BASELINE configurations: tools/ab/ab_env_cfg.sh SEGFAC_GEMM8_LINEAR cfg3 cfg5 "cfg5 --fp8"
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for c in "${@:-cfg3 cfg4 cfg5}"; do for v in 0 1 0 1; do
  if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
  python3 bench.py --config $c --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $VAR=$v', d['value'], d['ms_per_step'])"
done; done
