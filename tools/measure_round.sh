#!/bin/bash
# One gpurun call that produces the round's measurement artefacts under gpurun_out/m/ (copied into profiles/ afterwards):
#   tools/measure_round.sh <tag>            e.g. r02
# default bench line (with the CPU baseline), kernel stats of the same command, PMC traffic (two passes), small-batch lines,
# the other BASELINE configs.
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${TAG}_bench_b128.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_bench_b128_profiled.json 2>> $O/bench.err
python3 $R/tools/rocpd_summary.py $O/prof/stats_results.db --csv $O/${TAG}_bench_b128_kernel_stats.csv --top 12 > $O/stats_top.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --batch 128 --out $O/pmc_traffic_b128.json > $O/pmc_top.txt 2>&1
# the default line once more with the counter file just taken (same sources: bench.py reports roofline.traffic from it)
cp $O/pmc_traffic_b128.json $R/profiles/pmc_traffic_b128.json && python3 bench.py > $O/${TAG}_bench_b128_final.json 2>> $O/bench.err
for b in 4 16 32; do python3 bench.py --batch $b --no-cpu-baseline > $O/${TAG}_bench_b${b}.json 2>> $O/bench.err; done
python3 bench.py --config cfg3 --batch 32 --no-cpu-baseline > $O/${TAG}_bench_cfg3_b32.json 2>> $O/bench.err
python3 bench.py --config cfg4 --batch 16 --no-cpu-baseline > $O/${TAG}_bench_cfg4_b16.json 2>> $O/bench.err
python3 bench.py --config cfg5 --batch 8 --no-cpu-baseline > $O/${TAG}_bench_cfg5_b8.json 2>> $O/bench.err
python3 bench.py --config cfg5 --batch 8 --no-cpu-baseline --fp8 > $O/${TAG}_bench_cfg5_b8_fp8.json 2>> $O/bench.err
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
ls -la $O; cat $O/stats_top.txt | cut -c1-140; cat $O/pmc_top.txt | head -12; for f in $O/*bench*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('cpu_baseline',{}).get('value'))
except Exception as e: print('$f', 'ERR', e)"; done; tail -5 $O/bench.err
