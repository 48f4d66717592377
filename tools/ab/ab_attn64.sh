#!/bin/bash
# r05 head-dim-64 attention: parity of the new forward / query-side backward (prescaled and not), errors against fp32, the probe in
# one process (policy A/B) and against another build of the library, and the cfg4 step: tools/ab/ab_attn64.sh <other .so>
cd $GRAFT_REPO_ROOT
OTHER=$GRAFT_REPO_ROOT/segmentation_factory_amd/$1
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "test_attention" 2>&1 | tail -3
SEGFAC_ATTN64_PRESCALE=1 timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "test_attention" 2>&1 | tail -3
python3 tools/probe/attn64_accuracy.py 2>&1 | tail -2
ATTN_SPREAD=4 python3 tools/probe/attn64_accuracy.py 2>&1 | tail -2
SEGFAC_HIP_LIB=$OTHER PROBE_AB='attn_no_mfma=0' python3 tools/probe/attn64_accuracy.py 2>&1 | tail -1
echo "== probe new (prescale on / off alternating)"
PROBE_AB='attn64_prescale=0;attn64_prescale=1' python3 tools/probe/mfma_probe.py attn 10 2>/dev/null | tail -6
echo "== probe old"
SEGFAC_HIP_LIB=$OTHER python3 tools/probe/mfma_probe.py attn 10 2>/dev/null | tail -1
for v in new newpre old; do
  unset SEGFAC_HIP_LIB SEGFAC_ATTN64_PRESCALE
  if [ $v = old ]; then export SEGFAC_HIP_LIB=$OTHER; fi
  if [ $v = newpre ]; then export SEGFAC_ATTN64_PRESCALE=1; fi
  python3 bench.py --config cfg4 --batch 16 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $v', d['value'], d['ms_per_step'])"
done
