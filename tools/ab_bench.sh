#!/bin/bash
# same-box A/B of bench.py under environment switches: tools/ab_bench.sh "VAR1=1" "VAR2=1" ...   ("" = defaults); two rounds
for round in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "round $round [$cfg] $r"
  done
done
