# In-situ A/B of one environment setting: kernel traces of the same bench step with and without it, aligned dispatch by dispatch
# (tools/probe/trace_ab.py).  usage (GPU box): bash tools/probe/trace_ab.sh SEGFAC_GEMM8_LINEAR=0 cfg4 16 [--fp8]
cd /tmp && export TMPDIR=/tmp
SET=$1; cfg=$2; b=$3; shift; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tab; mkdir -p $O
for v in a b; do
  if [ $v = b ]; then export $SET; else unset ${SET%%=*}; fi
  rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_$v -o p -- python3 $R/bench.py --config $cfg --batch $b --no-cpu-baseline --no-extra-legs --steps 4 --warmup 2 "$@" > $O/bench_$v.json 2> $O/err_$v.txt
  f=$(find $O/prof_$v -name '*kernel_trace.csv' | head -1); cp $f $O/trace_$v.csv; rm -rf $O/prof_$v
done
unset ${SET%%=*}
python3 $R/tools/probe/trace_ab.py $O/trace_a.csv $O/trace_b.csv | tee $O/${cfg}_b${b}_trace_ab.txt
rm -f $O/trace_a.csv $O/trace_b.csv
