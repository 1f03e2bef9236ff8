set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/r3_smoke.log 2>&1; tail -1 $OUT/r3_smoke.log | cut -c1-200
for k in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3_gpu_tests_final_$k.log 2>&1; tail -1 $OUT/r3_gpu_tests_final_$k.log | cut -c1-200
done
timeout 300 python bench.py > $OUT/r3_bench_n1.json 2> $OUT/r3_bench_n1.err; cut -c1-300 $OUT/r3_bench_n1.json
rm -rf /tmp/prof_c3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > /dev/null 2>&1)
first=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_config3_resnet50_seethrough | head -16
