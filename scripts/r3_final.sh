# Final validation at HEAD: smoke, three consecutive full GPU suites, bench lines, rocprofv3 summary, kernel bench, all five
# BASELINE configurations in one process.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/r3_smoke.log 2>&1; tail -2 $OUT/r3_smoke.log | cut -c1-300
for k in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3_gpu_tests_final_$k.log 2>&1; tail -2 $OUT/r3_gpu_tests_final_$k.log | cut -c1-200
done
timeout 300 python bench.py > $OUT/r3_bench_n1.json 2> $OUT/r3_bench_n1.err; cut -c1-400 $OUT/r3_bench_n1.json
timeout 200 python bench.py --trials-per-gpu 4 --cpu-baseline-iters 0 --no-dry-collective > $OUT/r3_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-200 $OUT/r3_bench_n1_4trials_in_flight.json
timeout 300 python bench.py --gpus 2 --steps 50 --cpu-baseline-iters 0 2>/dev/null | grep '^{"metric' > $OUT/r3_bench_2ranks_one_gpu.json; cut -c1-200 $OUT/r3_bench_2ranks_one_gpu.json
rm -rf /tmp/prof_bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-dry-collective > $OUT/r3_bench_under_rocprof.json 2> /dev/null)
first=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_bench | head -14
cp $(dirname $first)/*kernel_stats.csv $OUT/r3_bench_rocprofv3_kernel_stats.csv
timeout 400 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -1 $OUT/r3_kernel_bench.err
rm -rf /tmp/prof_c3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > /dev/null 2>&1)
first=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_config3_resnet50_seethrough | head -16
timeout 900 python scripts/config_runs.py --full > $OUT/r3_config_runs_same_process.log 2>&1; grep "configs\[" $OUT/r3_config_runs_same_process.log | grep iterations | cut -c1-260
