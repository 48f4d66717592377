#!/bin/bash
# same-box on / off of one environment switch on the cfg2 step at several batches: tools/ab/ab_env.sh SEGFAC_DW_NO_SMALL [batches...]
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for b in ${@:-4 16 128}; do for v in 0 1; do
  if [ $v = 1 ]; then export $VAR=1; else unset $VAR; fi
  python3 bench.py --batch $b --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b $VAR=$v', d['value'], d['ms_per_step'])"
done; done
