# fabric fetch / L2 hit / time of the loss kernels at the cfg2 batch (tools/probe/loss_probe.py): bash tools/probe/pmc_lossfwd.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; rm -f $O/summary.txt
python3 $R/tools/probe/loss_probe.py 1 2>&1 | tail -1 > $O/summary.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/probe/loss_probe.py 1 > $O/p$i.log 2>&1
  python3 - $O/p$i >> $O/summary.txt <<'PY'
import csv, sys, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'band' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], re.sub(r'\(.*', '', r['Kernel_Name'])[:45], r['Counter_Name'])] += float(r['Counter_Value'])
    for (_, k, c), v in per.items():
        acc[k][c].append(v)
for k, v in sorted(acc.items()):
    print(k, ' | '.join(f'{c} {sum(x) / len(x):.4g}' for c, x in sorted(v.items())))
PY
  rm -rf $O/p$i
done
cat $O/summary.txt
