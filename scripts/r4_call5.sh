# Round 4, GPU call 5: the whole GPU suite after the lazy-BatchNorm / DeepInversion fix; rocprofv3 --stats of the bench command.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 2400 python -m pytest tests -m gpu -q --durations=15 > $OUT/r4_gpu_tests_call5.log 2>&1; tail -40 $OUT/r4_gpu_tests_call5.log | cut -c1-250
rm -rf /tmp/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-hbm-resident > $OUT/r4_bench_under_rocprof.json 2> $OUT/r4_bench_under_rocprof.err)
first=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
if [ -n "$first" ]; then
  python scripts/summarize_prof.py $(dirname $first) $OUT/r4_bench | head -16
  ls $(dirname $first)
  stats=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$stats" ] && cp "$stats" $OUT/r4_bench_rocprofv3_kernel_stats.csv
fi
