# PMC counter sets over one short bench run; per-kernel averages for the heaviest kernels (whole chip sums per launch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ps; mkdir -p $O; rm -f $O/summary.txt
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 2 --warmup 1 > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $O/summary.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    for key in ('bn_cls_bwd_kernel<1','bn_cls_bwd_kernel<2','fuse_map_kernel','fuse_map_bwd','gemm_bf16_big_kernel<0, unsigned short, false, true','attn_mfma_bwd_fused','dwconv3x3_walk_kernel<unsigned short, 2>','ln_bwd_kernel','gemm_skinny_k_kernel<1, 6, 2>','ce_dice_fwd_band'):
        if key in k: acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    for c,vals in v.items(): print(f'{k:60s} {c:28s} {sum(vals)/len(vals):16.0f}  n={len(vals)}')
PY
  fi
  rm -rf $O/p$i
done
sort $O/summary.txt
