set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/r3_kernel_tests_producer_stats.log 2>&1; tail -8 $OUT/r3_kernel_tests_producer_stats.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_attack.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "resnet50 or legacy or seethrough or resnet18_imagenet or convnet" > $OUT/r3_e2e_producer_stats.log 2>&1; tail -4 $OUT/r3_e2e_producer_stats.log | cut -c1-300
R=$OUT/r3_producer_stats_ab.txt
: > $R
for P in 1 0; do
  rm -rf /tmp/prof_p$P
  (cd /tmp && BREACH_HIP_BN_PRODUCER_STATS=$P timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_p$P -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 --its 80 > /dev/null 2>&1)
  first=$(find /tmp/prof_p$P -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_prof.py $(dirname $first) /tmp/sum_p$P > /dev/null
  echo "== BREACH_HIP_BN_PRODUCER_STATS=$P" >> $R
  grep "bn_sums\|bn_finalize\|bn_eval_fwd" /tmp/sum_p${P}_kernel_summary.csv | cut -c1-170 >> $R
  BREACH_HIP_BN_PRODUCER_STATS=$P timeout 200 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[2\]" | head -1 | cut -c88-200 >> $R
done
cat $R
