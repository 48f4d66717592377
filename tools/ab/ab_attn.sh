#!/bin/bash
# same-box A/B of two builds of the library on the head-dim-64 attention probe and the cfg4 step: tools/ab/ab_attn.sh <other .so>
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/segmentation_factory_amd/$1
for r in 1 2; do
for v in new old; do
  if [ $v = old ]; then export SEGFAC_HIP_LIB=$OTHER; else unset SEGFAC_HIP_LIB; fi
  echo "== $v"; python3 tools/probe/mfma_probe.py attn 10 2>/dev/null | tail -2
  python3 bench.py --config cfg4 --batch 16 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['value'], d['ms_per_step'])"
done; done
