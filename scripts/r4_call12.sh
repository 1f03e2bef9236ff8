# Round 4, GPU call 12: rocprofv3 --kernel-trace --stats of the bench command at HEAD (kernel summary, stats, gap census).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 300 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 100 > /dev/null 2>&1   # MIOpen's solver search happens here, not under the profiler
rm -rf /tmp/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-hbm-resident --no-dry-collective > $OUT/r4_bench_under_rocprof.json 2> $OUT/r4_bench_under_rocprof.err)
trace=$(ls -S $(find /tmp/prof_bench -name "*kernel_trace.csv") | head -1)
if [ -n "$trace" ]; then
  python scripts/summarize_prof.py $(dirname $trace) $OUT/r4_bench | head -16
  stats=$(ls -S $(find /tmp/prof_bench -name "*kernel_stats.csv") | head -1); [ -n "$stats" ] && cp "$stats" $OUT/r4_bench_rocprofv3_kernel_stats.csv
  python scripts/gap_census.py $trace $OUT/r4_1trial_head_gap_census --iters 60 --skip-tail 45 --label "1 trial at HEAD (kernel E epilogue + folded accumulation)" | head -28
fi
cut -c1-200 $OUT/r4_bench_under_rocprof.json
