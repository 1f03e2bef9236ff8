# Round 4, GPU call 2: new tests (fused DeepInversion tap, ABI 5, concurrency, step direction), kernel D backward A/B profile,
# memory-copy trace around the large in-graph gaps, launch attribution, stream -> pipe probe.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/r4_smoke.log 2>&1; tail -1 $OUT/r4_smoke.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/r4_gpu_tests_call2.log 2>&1; tail -30 $OUT/r4_gpu_tests_call2.log | cut -c1-220
for mode in 1 0; do
  rm -rf /tmp/prof_c3
  (cd /tmp && BREACH_HIP_BN_FUSED_TAP=$mode timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > $OUT/r4_config3_tap$mode.log 2>&1)
  first=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_prof.py $(dirname $first) $OUT/r4_config3_fused_tap_$mode | head -16
  grep "configs\[" $OUT/r4_config3_tap$mode.log | cut -c1-200
done
for mode in 1 0; do BREACH_HIP_BN_FUSED_TAP=$mode timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | cut -c1-200 | tee -a $OUT/r4_config3_fused_tap_noprof.log; done
rm -rf /tmp/prof_mc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_mc -- python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-iters 0 --no-dry-collective --steps 60 --warmup 20 --no-kernel-timing --no-span-timing > $OUT/r4_memcopy_stdout.log 2> $OUT/r4_memcopy_stderr.log)
find /tmp/prof_mc -name "*.csv" | head; mc=$(find /tmp/prof_mc -name "*memory_copy_trace.csv" | head -1); [ -n "$mc" ] && (wc -l $mc; head -5 $mc; cp $mc $OUT/r4_memory_copy_trace.csv)
timeout 400 python scripts/op_attribution.py > $OUT/r4_op_attribution.json 2> $OUT/r4_op_attribution.txt; head -70 $OUT/r4_op_attribution.txt
timeout 900 python scripts/inflight_pipes_probe.py --sweep > $OUT/r4_inflight_pipes_probe.jsonl 2> $OUT/r4_inflight_pipes_probe.err; cut -c1-220 $OUT/r4_inflight_pipes_probe.jsonl
