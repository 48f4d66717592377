#!/bin/bash
# same-box A/B of two builds of the library on a probe: tools/ab/ab_probe.sh <other .so under segmentation_factory_amd/> <probe args...>
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/segmentation_factory_amd/$1; shift
for r in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export SEGFAC_HIP_LIB=$OTHER; else unset SEGFAC_HIP_LIB; fi
  echo "== $v"; python3 tools/probe/mfma_probe.py "$@" 2>/dev/null | grep -E "conv3x3|attention"
done; done
