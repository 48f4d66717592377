# Counters that decide between "schedule" and "clock" for the eight-phase GEMM: per launch of the UPerHead bottleneck
# 3x3 conv (tools/probe/mfma_probe.py conv) the held clock (GRBM_GUI_ACTIVE / 8 / time), MFMA-busy, wave-parked share, L2 hit rate and
# fabric fetch -- under each dispatch-policy setting given (csrc/policy.h), one rocprofv3 pass per counter set.
# usage (GPU box): bash tools/probe/pmc_g8.sh <tag> "SEGFAC_G8_FWD_ORDER=0" "SEGFAC_G8_FWD_ORDER=1" ...
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; rm -f $O/summary.txt
for setting in "$@"; do
  export $setting
  echo "== $setting" >> $O/summary.txt
  python3 $R/tools/probe/mfma_probe.py conv 5 2>&1 | grep conv3x3 >> $O/summary.txt
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/probe/mfma_probe.py conv 2 > $O/p$i.log 2>&1
    python3 - $O/p$i >> $O/summary.txt <<'PY'
import csv, sys, glob, collections, re
d = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm8_kernel' in r['Kernel_Name']:
            dur[re.sub(r'\(.*', '', r['Kernel_Name'])[:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'gemm8_kernel' in r['Kernel_Name']:
            per[(r['Dispatch_Id'], re.sub(r'\(.*', '', r['Kernel_Name'])[:60], r['Counter_Name'])] += float(r['Counter_Value'])
    for (_, k, c), v in per.items():
        acc[k][c].append(v)
for k, v in sorted(acc.items()):
    ms = sum(dur[k]) / max(len(dur[k]), 1)
    line = f'{k:62s} {ms:8.3f} ms(profiled)'
    for c, vals in sorted(v.items()):
        a = sum(vals) / len(vals)
        line += f' | {c} {a:.4g}'
        if c == 'GRBM_GUI_ACTIVE' and ms > 0:
            line += f' (clock {a / 8 / (ms * 1e-3) / 1e9:.2f} GHz)'
    if 'TCC_HIT_sum' in v:
        h, m = sum(v['TCC_HIT_sum']), sum(v['TCC_MISS_sum'])
        line += f' | L2 hit {100 * h / (h + m):.1f} %'
    if 'FETCH_SIZE' in v:
        line += f" | fabric fetch {2 * 1024 * sum(v['FETCH_SIZE']) / len(v['FETCH_SIZE']) / 1e9:.2f} GB per launch (x2 applied)"
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v:
        line += f" | MFMA-busy {100 * sum(v['SQ_VALU_MFMA_BUSY_CYCLES']) / (sum(v['GRBM_GUI_ACTIVE']) / 8 * 1024):.1f} %"
    print(line)
PY
    rm -rf $O/p$i
  done
  unset ${setting%%=*}
done
cat $O/summary.txt
