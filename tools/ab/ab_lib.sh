#!/bin/bash
# same-box A/B of two builds of the library on the default bench line: tools/ab/ab_lib.sh <other .so under segmentation_factory_amd/> [bench args]
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/segmentation_factory_amd/$1; shift
for r in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export SEGFAC_HIP_LIB=$OTHER; else unset SEGFAC_HIP_LIB; fi
  python3 bench.py --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
done; done
