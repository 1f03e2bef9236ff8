# Round 4, GPU call 6: the four tests that failed in call 5 (thresholds / yardsticks corrected), determinism probe, rocprofv3 --stats of the bench command.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 900 python -m pytest tests -m gpu -q -s -k "side_streams or share_priors or langevin_noise_under or 32_restarts or in_flight_match or graph_replay_and_eager" > $OUT/r4_gpu_tests_call6.log 2>&1; tail -30 $OUT/r4_gpu_tests_call6.log | cut -c1-250
timeout 600 python scripts/determinism_probe.py > $OUT/r4_determinism_probe.jsonl 2> $OUT/r4_determinism_probe.err; cut -c1-250 $OUT/r4_determinism_probe.jsonl; tail -3 $OUT/r4_determinism_probe.err
rm -rf /tmp/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-hbm-resident --no-dry-collective > $OUT/r4_bench_under_rocprof.json 2> $OUT/r4_bench_under_rocprof.err)
trace=$(ls -S $(find /tmp/prof_bench -name "*kernel_trace.csv") | head -1)
if [ -n "$trace" ]; then
  python scripts/summarize_prof.py $(dirname $trace) $OUT/r4_bench | head -16
  stats=$(ls -S $(find /tmp/prof_bench -name "*kernel_stats.csv") | head -1); [ -n "$stats" ] && cp "$stats" $OUT/r4_bench_rocprofv3_kernel_stats.csv
fi
cut -c1-200 $OUT/r4_bench_under_rocprof.json
