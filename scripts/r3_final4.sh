set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/r3_smoke.log 2>&1; tail -1 $OUT/r3_smoke.log | cut -c1-200
for k in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3_gpu_tests_final_$k.log 2>&1; tail -1 $OUT/r3_gpu_tests_final_$k.log | cut -c1-200
done
rm -rf /tmp/prof_c5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5 > /dev/null 2>&1)
first=$(find /tmp/prof_c5 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_config5_bert_tag | head -14
timeout 900 python scripts/config_runs.py --full --its 1000 > $OUT/r3_config_runs_same_process.log 2>&1; grep "configs\[" $OUT/r3_config_runs_same_process.log | grep iterations | cut -c1-230
