# PMC counter sets over the MFMA-bound probe kernels (tools/probe/mfma_probe.py): per-kernel per-launch averages, whole-chip sums.
# usage (on the GPU box): bash tools/probe/pmc_mfma.sh [conv|attn|all] [tag]
cd /tmp && export TMPDIR=/tmp
WHAT=${1:-all}; TAG=${2:-pm}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; rm -f $O/summary.txt
python3 $R/tools/probe/mfma_probe.py $WHAT 5 > $O/time.txt 2>&1
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/tools/probe/mfma_probe.py $WHAT 2 > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $O/summary.txt <<'PY'
import csv,sys,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'gemm8_kernel' in k or 'attn_' in k:
        k=re.sub(r'\(.*','',k)[:70]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items(): print(f'{k:72s} {c:28s} {sum(vals)/len(vals):18.0f}  n={len(vals)}')
PY
  else tail -3 $O/p$i.log >> $O/summary.txt; fi
  rm -rf $O/p$i
done
cat $O/time.txt; sort $O/summary.txt
