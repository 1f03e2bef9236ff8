# Profiles behind DESIGN.md section 6 (run on the GPU box through gpurun; summaries are copied to profiles/ by hand).
# Every step runs under its own `timeout`: a hung profiler pass must not eat the GPU budget.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
prof() {  # prof <seconds> <tag> <pmc-counter or ""> <command...>
  limit=$1; tag=$2; counter=$3; shift 3
  rm -rf /tmp/prof_$tag
  if [ -z "$counter" ]; then
    (cd /tmp && timeout $limit rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  else
    (cd /tmp && timeout $limit rocprofv3 --pmc $counter --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  fi
  first=$(find /tmp/prof_$tag -name "*.csv" | head -1)
  if [ -n "$first" ]; then
    dir=$(dirname $first)
    python scripts/summarize_prof.py $dir $OUT/$tag $counter | head -16
    cp $dir/*kernel_stats.csv $OUT/${tag}_rocprofv3_kernel_stats.csv 2>/dev/null
  fi
  tail -1 $OUT/${tag}_stdout.log | cut -c1-300
}
B="python $GRAFT_REPO_ROOT/bench.py"
prof 300 r2_bench "" $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing
prof 300 r2_config3_resnet50_seethrough "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3
prof 300 r2_config5_bert_tag "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5
prof 300 r2_bench_4trials "" $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --trials-per-gpu 4 --no-span-timing
timeout 300 $B > $OUT/r2_bench_n1.json 2>/dev/null; cut -c1-250 $OUT/r2_bench_n1.json
timeout 200 $B --trials-per-gpu 4 --cpu-baseline-iters 0 > $OUT/r2_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-200 $OUT/r2_bench_n1_4trials_in_flight.json
timeout 200 $B --miopen-benchmark --cpu-baseline-iters 0 > $OUT/r2_bench_miopen_benchmark.json 2>/dev/null; cut -c1-200 $OUT/r2_bench_miopen_benchmark.json
timeout 200 $B --channels-last --cpu-baseline-iters 0 > $OUT/r2_bench_channels_last.json 2>/dev/null; cut -c1-200 $OUT/r2_bench_channels_last.json
timeout 300 python scripts/kernel_bench.py > $OUT/r2_kernel_bench.json 2> $OUT/r2_kernel_bench.err; tail -2 $OUT/r2_kernel_bench.err
timeout 300 $B --gpus 2 --steps 50 --cpu-baseline-iters 0 > $OUT/r2_bench_2ranks_one_gpu.json 2> $OUT/r2_bench_2ranks_one_gpu.err; cut -c1-300 $OUT/r2_bench_2ranks_one_gpu.json; tail -3 $OUT/r2_bench_2ranks_one_gpu.err
timeout 900 python scripts/config_runs.py --full > $OUT/r2_config_runs.log 2>&1; tail -30 $OUT/r2_config_runs.log
prof 200 r2_pmc_fetch FETCH_SIZE $B --steps 10 --warmup 4 --no-graph --cpu-baseline-iters 0 --no-kernel-timing --no-span-timing
prof 200 r2_pmc_write WRITE_SIZE $B --steps 10 --warmup 4 --no-graph --cpu-baseline-iters 0 --no-kernel-timing --no-span-timing
