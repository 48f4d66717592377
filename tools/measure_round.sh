#!/bin/bash
# One gpurun call that produces the round's measurement artefacts under gpurun_out/m/ (copied into profiles/ afterwards):
#   tools/measure_round.sh <tag> [sections]     e.g. r03 "main small cfgs parity" (default: all five)
# default bench line (with the CPU baseline), kernel stats of the same command, PMC traffic (two passes), small-batch lines,
# the other BASELINE configs.  Run it as the LAST act of a round, on the committed sources: the counter file it leaves in
# profiles/pmc_traffic_b${HB}.json is stamped with one hash per reported kernel (bench.KERNEL_SOURCES) and bench.py reports a kernel's
# traffic only while ITS sources are unchanged (r04's closing commit touched gemm.hip and voided the loss kernel's figure too).
set -u
HB=256          # the headline per-GPU batch (bench.py default)
TAG=${1:-r05}
SECTIONS=${2:-main small cfgs power parity check}
has() { case " $SECTIONS " in *" $1 "*) return 0;; *) return 1;; esac; }
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if has main; then
python3 $R/bench.py > $O/${TAG}_bench_b${HB}.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o stats -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 10 --warmup 3 > $O/${TAG}_bench_b${HB}_profiled.json 2>> $O/bench.err
python3 $R/tools/rocpd_summary.py $O/prof/stats_results.db --csv $O/${TAG}_bench_b${HB}_kernel_stats.csv --top 12 > $O/stats_top.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o fetch -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o write -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --batch $HB --out $O/pmc_traffic_b${HB}.json > $O/pmc_top.txt 2>&1
# the default line once more with the counter file just taken (same sources: bench.py reports roofline.traffic from it)
cp $O/pmc_traffic_b${HB}.json $R/profiles/pmc_traffic_b${HB}.json && python3 bench.py > $O/${TAG}_bench_b${HB}_final.json 2>> $O/bench.err
fi
if has small; then
cd $R
for b in 4 16 32 128; do python3 bench.py --batch $b --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_b${b}.json 2>> $O/bench.err; done
fi
if has cfgs; then
cd $R
python3 bench.py --config cfg3 --batch 64 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_cfg3_b64.json 2>> $O/bench.err
python3 bench.py --config cfg4 --batch 32 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_cfg4_b32.json 2>> $O/bench.err
python3 bench.py --config cfg5 --batch 32 --no-cpu-baseline --no-extra-legs > $O/${TAG}_bench_cfg5_b32.json 2>> $O/bench.err
python3 bench.py --config cfg5 --batch 32 --no-cpu-baseline --no-extra-legs --fp8 > $O/${TAG}_bench_cfg5_b32_fp8.json 2>> $O/bench.err
python3 bench.py --config cfg3 --batch 64 --no-cpu-baseline --no-extra-legs --fp8 > $O/${TAG}_bench_cfg3_b64_fp8.json 2>> $O/bench.err
python3 tools/probe/fp8_conv_probe.py > $O/${TAG}_conv3x3_bf16_vs_fp8.txt 2>> $O/bench.err
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
# kernel statistics + counter traffic of the other BASELINE configs (the TFLOP/s and GB/s claims of DESIGN section 5 / 9)
for cb in cfg3:64 cfg4:32 cfg5:32; do
  c=${cb%%:*}; b=${cb##*:}
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof_$c -o stats -- python3 $R/bench.py --config $c --batch $b --no-cpu-baseline --no-extra-legs --steps 5 --warmup 2 > /dev/null 2>> $O/bench.err
  python3 $R/tools/rocpd_summary.py $O/prof_$c/stats_results.db --csv $O/${TAG}_${c}_b${b}_kernel_stats.csv --top 10 > $O/stats_top_$c.txt 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$c -o fetch -- python3 $R/bench.py --config $c --batch $b --no-cpu-baseline --no-extra-legs --steps 1 --warmup 1 > /dev/null 2>> $O/bench.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$c -o write -- python3 $R/bench.py --config $c --batch $b --no-cpu-baseline --no-extra-legs --steps 1 --warmup 1 > /dev/null 2>> $O/bench.err
  cd $R && python3 tools/pmc_traffic.py $O/pmcf_$c $O/pmcw_$c --batch $b --out $O/${TAG}_pmc_traffic_${c}_b${b}.json > $O/pmc_top_$c.txt 2>&1
  rm -rf $O/prof_$c $O/pmcf_$c $O/pmcw_$c
done
fi
if has power; then
cd $R
# socket power and shader clock while each configuration's step replays (rocm-smi, one sample per second; r05: the MFMA-dense
# configurations run at the package power limit and the clock the firmware grants, not at 2.4 GHz)
P=$O/${TAG}_power_clock.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" > $P
for cbs in cfg2:$HB:200 cfg3:64:60 cfg4:32:60 cfg5:32:40; do
  c=${cbs%%:*}; rest=${cbs#*:}; b=${rest%%:*}; n=${rest##*:}
  echo "== $c batch $b" >> $P
  touch $O/.sampling
  ( while [ -e $O/.sampling ]; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed -e 's/.*sclk clock level: [0-9]*: (\([0-9]*\)Mhz).*/sclk \1 MHz/' -e 's/.*Power (W): \([0-9.]*\).*/power \1 W/' | tr "\n" " "; echo; sleep 1; done ) >> $P &
  python3 bench.py --config $c --batch $b --no-cpu-baseline --no-extra-legs --steps $n 2>> $O/bench.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms/step over', d['steps'], 'steps')" > $O/.bline
  rm -f $O/.sampling; wait
  cat $O/.bline >> $P; rm -f $O/.bline
done
python3 - $P <<'PY' >> $P
import re, sys
cur, rows, rate = None, {}, {}
for ln in open(sys.argv[1]):
    if ln.startswith('== '):
        cur = ln[3:].strip(); rows[cur] = []
    m = re.search(r'sclk (\d+) MHz.*power ([\d.]+) W', ln)
    if m and cur:
        rows[cur].append((int(m.group(1)), float(m.group(2))))
    m = re.match(r'bench ([0-9.]+) images/sec', ln)
    if m and cur:
        rate[cur] = float(m.group(1))
print('-- summary (samples at >= 70 % of the largest power seen in the run = the replay phase)')
for k, v in rows.items():
    if not v:
        continue
    top = max(p for _, p in v)
    hot = [(c, p) for c, p in v if p >= 0.7 * top]
    mp = sum(p for _, p in hot) / len(hot)
    print(f'{k}: {len(hot)} samples, mean power {mp:.0f} W, mean sclk {sum(c for c, _ in hot) / len(hot):.0f} MHz (min {min(c for c, _ in hot)}, max {max(c for c, _ in hot)})'
          + (f', {mp / rate[k]:.2f} J per image' if k in rate else ''))
PY
fi
if has parity; then
cd $R
# per-tensor gradient errors at full size behind the bf16 bars of test_full_size_fp32_and_bf16_vs_oracle
python3 tools/grad_parity_fullsize.py cfg2 cfg3 cfg4 cfg5 > $O/${TAG}_grad_parity_fullsize.txt 2>> $O/bench.err
fi
if has check; then
cd $R
# index widths at the benchmarked batches: a 2n-image batch against its n-image halves, eval mode (tools/check_large_batch.py)
for nc in 128:cfg2 32:cfg3 16:cfg4 16:cfg5; do timeout 400 python3 tools/check_large_batch.py ${nc%%:*} ${nc##*:} 2>&1 | tail -2; done > $O/${TAG}_large_batch_check.txt
fi
ls -la $O; for f in $O/stats_top*.txt; do echo "== $f"; cut -c1-150 $f; done; for f in $O/pmc_top*.txt; do echo "== $f"; head -12 $f | cut -c1-150; done; head -60 $O/*grad_parity_fullsize.txt 2>/dev/null; for f in $O/*bench*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('cpu_baseline',{}).get('value'))
except Exception as e: print('$f', 'ERR', e)"; done; tail -5 $O/bench.err
