# Profiles behind DESIGN.md section 6 (run on the GPU box through gpurun; summaries are copied to profiles/ by hand).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
prof() {  # prof <tag> <pmc-counter or ""> <command...>
  tag=$1; counter=$2; shift 2
  rm -rf /tmp/prof_$tag
  if [ -z "$counter" ]; then
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  else
    (cd /tmp && rocprofv3 --pmc $counter --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  fi
  dir=$(dirname $(find /tmp/prof_$tag -name "*.csv" | head -1))
  python scripts/summarize_prof.py $dir $OUT/$tag $counter | head -30
  cp $dir/*kernel_stats.csv $OUT/${tag}_rocprofv3_kernel_stats.csv 2>/dev/null
  tail -2 $OUT/${tag}_stdout.log | cut -c1-600
}
B="python $GRAFT_REPO_ROOT/bench.py"
prof r2_bench "" $B --steps 100 --warmup 20 --cpu-baseline-iters 0
prof r2_pmc_fetch FETCH_SIZE $B --steps 10 --warmup 4 --no-graph --cpu-baseline-iters 0 --no-kernel-timing
prof r2_pmc_write WRITE_SIZE $B --steps 10 --warmup 4 --no-graph --cpu-baseline-iters 0 --no-kernel-timing
prof r2_bench_4trials "" $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --trials-per-gpu 4
prof r2_config3_resnet50_seethrough "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3
prof r2_config5_bert_tag "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5
$B --trials-per-gpu 4 --cpu-baseline-iters 0 > $OUT/r2_bench_n1_4trials_in_flight.json 2>/dev/null; cat $OUT/r2_bench_n1_4trials_in_flight.json | cut -c1-300
$B --miopen-benchmark --cpu-baseline-iters 0 > $OUT/r2_bench_miopen_benchmark.json 2>/dev/null; cut -c1-200 $OUT/r2_bench_miopen_benchmark.json
$B --channels-last --cpu-baseline-iters 0 > $OUT/r2_bench_channels_last.json 2>/dev/null; cut -c1-200 $OUT/r2_bench_channels_last.json
