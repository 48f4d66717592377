cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pc; mkdir -p $O
cfg=$1; b=$2; shift; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --config $cfg --batch $b --no-cpu-baseline --no-extra-legs --steps 5 --warmup 2 "$@" > $O/bench_$cfg.json 2> $O/err.txt
f=$(ls $O/prof/*kernel_stats.csv | head -1); cp $f $O/${cfg}_kernel_stats.csv
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={100*float(r['TotalDurationNs'])/tot:5.2f}")
PY
tail -c 600 $O/bench_$cfg.json | head -c 300
rm -rf $O/prof
