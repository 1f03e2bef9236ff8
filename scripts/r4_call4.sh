# Round 4, GPU call 4: the whole GPU suite, profiles of the round (rocprofv3 stats of the bench command, PMC passes, kernel D backward A/B),
# final-form bench lines, all BASELINE configurations in one process.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 1800 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/r4_gpu_tests_call4.log 2>&1; tail -22 $OUT/r4_gpu_tests_call4.log | cut -c1-250
pmc() {  # pmc <seconds> <tag> <counter list> <command...>
  limit=$1; tag=$2; counters=$3; shift 3
  rm -rf /tmp/prof_$tag
  (cd /tmp && timeout $limit rocprofv3 --pmc $counters --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  first=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$first" ]; then python scripts/summarize_prof.py $(dirname $first) $OUT/$tag | head -8; fi
}
T="python $GRAFT_REPO_ROOT/scripts/pmc_target.py"
pmc 150 r4_pmc_fetch_resnet18 FETCH_SIZE $T --size resnet18
pmc 150 r4_pmc_write_resnet18 WRITE_SIZE $T --size resnet18
for v in plain tap; do
  pmc 150 r4_pmc_fetch_bneval_$v FETCH_SIZE $T --size bneval_$v
  pmc 150 r4_pmc_write_bneval_$v WRITE_SIZE $T --size bneval_$v
done
rm -rf /tmp/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-hbm-resident > $OUT/r4_bench_under_rocprof.json 2> $OUT/r4_bench_under_rocprof.err)
first=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
if [ -n "$first" ]; then python scripts/summarize_prof.py $(dirname $first) $OUT/r4_bench | head -24; cp $(dirname $first)/*kernel_stats.csv $OUT/r4_bench_rocprofv3_kernel_stats.csv; fi
timeout 300 python scripts/config_runs.py --only 3 > /dev/null 2>&1   # MIOpen solver search for ResNet-50 B = 8 happens here, not inside an A/B leg
for mode in 1 0; do
  rm -rf /tmp/prof_c3
  (cd /tmp && BREACH_HIP_BN_FUSED_TAP=$mode timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > $OUT/r4_config3_tap$mode.log 2>&1)
  first=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_prof.py $(dirname $first) $OUT/r4_config3_fused_tap_$mode | head -8
  grep "configs\[" $OUT/r4_config3_tap$mode.log | head -1 | cut -c1-200
done
for mode in 1 0 1 0; do BREACH_HIP_BN_FUSED_TAP=$mode timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | head -1 | cut -c1-200 | sed "s/^/fused_tap=$mode /" | tee -a $OUT/r4_config3_fused_tap_noprof.log; done
timeout 600 $B > $OUT/r4_bench_n1.json 2> $OUT/r4_bench_n1.err; cut -c1-300 $OUT/r4_bench_n1.json
timeout 300 $B --trials-per-gpu 4 --cpu-baseline-iters 0 --no-hbm-resident > $OUT/r4_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-200 $OUT/r4_bench_n1_4trials_in_flight.json
timeout 400 $B --gpus 2 --steps 50 --cpu-baseline-iters 0 > $OUT/r4_bench_2ranks_one_gpu.json 2> $OUT/r4_bench_2ranks_one_gpu.err; cut -c1-300 $OUT/r4_bench_2ranks_one_gpu.json; tail -3 $OUT/r4_bench_2ranks_one_gpu.err | cut -c1-300
timeout 1200 python scripts/config_runs.py --full --its 1000 > $OUT/r4_config_runs_same_process.log 2>&1; grep "configs\[" $OUT/r4_config_runs_same_process.log | grep iterations | cut -c1-230
