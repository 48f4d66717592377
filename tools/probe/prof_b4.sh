cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pb; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --batch 4 --no-cpu-baseline --no-extra-legs --steps 20 --warmup 3 > $O/bench_b4.json 2> $O/err.txt
f=$(ls $O/prof/*kernel_stats.csv | head -1); cp $f $O/b4_kernel_stats.csv
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms', tot/1e6, 'kernels', len(rows))
for r in rows[:40]:
    print(f"{r['Name'][:95]:95s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={100*float(r['TotalDurationNs'])/tot:5.2f}")
PY
rm -rf $O/prof
