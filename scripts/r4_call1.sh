# Round 4, GPU call 1: gap census of the replayed iteration (1 / 4 / 6 trials in flight), per-node cost probe without a profiler,
# MIOpen solver-family A/B on the bench line.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-iters 0 --no-dry-collective"
trace() {  # trace <seconds> <tag> <command...>
  limit=$1; tag=$2; shift 2
  rm -rf /tmp/prof_$tag
  (cd /tmp && timeout $limit rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  first=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  if [ -n "$first" ]; then
    python scripts/gap_census.py $first $OUT/${tag}_gap_census --iters 60 --label "$tag" | head -40
    python scripts/summarize_prof.py $(dirname $first) $OUT/$tag > /dev/null
  fi
  tail -1 $OUT/${tag}_stdout.log | cut -c1-200
}
timeout 200 $B --steps 100 --warmup 20 > $OUT/r4_ab_default_first.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_default_first.json
trace 300 r4_1trial $B --steps 100 --warmup 20 --no-kernel-timing --no-span-timing
trace 300 r4_4trials $B --steps 60 --warmup 20 --no-kernel-timing --no-span-timing --trials-per-gpu 4
trace 300 r4_6trials $B --steps 40 --warmup 20 --no-kernel-timing --no-span-timing --trials-per-gpu 6
timeout 300 python scripts/node_cost_probe.py > $OUT/r4_node_cost_probe.jsonl 2> $OUT/r4_node_cost_probe.err; cat $OUT/r4_node_cost_probe.jsonl; tail -3 $OUT/r4_node_cost_probe.err
timeout 200 $B --steps 200 --warmup 20 > $OUT/r4_ab_default.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_default.json
MIOPEN_DEBUG_CONV_GEMM=0 timeout 300 $B --steps 200 --warmup 20 > $OUT/r4_ab_gemm0.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_gemm0.json
MIOPEN_DEBUG_CONV_GEMM=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 timeout 300 $B --steps 200 --warmup 20 > $OUT/r4_ab_gemm0_igemm0.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_gemm0_igemm0.json
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 timeout 300 $B --steps 200 --warmup 20 > $OUT/r4_ab_igemm0.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_igemm0.json
MIOPEN_DEBUG_CONV_GEMM=0 timeout 300 $B --steps 200 --warmup 20 --trials-per-gpu 4 > $OUT/r4_ab_gemm0_4trials.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_gemm0_4trials.json
timeout 200 $B --steps 200 --warmup 20 > $OUT/r4_ab_default_again.json 2>/dev/null; cut -c1-160 $OUT/r4_ab_default_again.json
MIOPEN_DEBUG_CONV_GEMM=0 trace 300 r4_1trial_gemm0 $B --steps 60 --warmup 20 --no-kernel-timing --no-span-timing
