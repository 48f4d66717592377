#!/bin/bash
# same-box on / off of the few-tile split-K form of the 3x3 convolution's forward / data gradient on cfg3 / cfg5 (r04 also toggled the
# weight gradient's tile rule through SEGFAC_CONV_WGRAD_OLD_RULE; that rule won and its switch is gone, csrc/policy.h)
cd $GRAFT_REPO_ROOT
for cb in cfg3:32 cfg5:8; do c=${cb%%:*}; b=${cb##*:}
for r in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export SEGFAC_CONV_NO_FWD_SPLIT=1; else unset SEGFAC_CONV_NO_FWD_SPLIT; fi
  python3 bench.py --config $c --batch $b --no-cpu-baseline --no-extra-legs $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $1 old=$v', d['value'], d['ms_per_step'])"
done; done; done
