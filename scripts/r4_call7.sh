# Round 4, GPU call 7: the 24 000-iteration tests, the whole suite with its printed evidence, eight full-length starts, final bench lines.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 900 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "24k or step_direction" > $OUT/r4_gpu_tests_24k.log 2>&1; tail -60 $OUT/r4_gpu_tests_24k.log | cut -c1-260
timeout 2400 python -m pytest tests -m gpu -q -s --durations=15 > $OUT/r4_gpu_tests.log 2>&1; tail -25 $OUT/r4_gpu_tests.log | cut -c1-200
timeout 900 python scripts/config_runs.py --only 24k --starts 8 > $OUT/r4_config1_24k_8starts.log 2>&1; tail -5 $OUT/r4_config1_24k_8starts.log | cut -c1-400
timeout 600 $B > $OUT/r4_bench_n1.json 2> $OUT/r4_bench_n1.err; cut -c1-300 $OUT/r4_bench_n1.json
timeout 300 $B --trials-per-gpu 4 --cpu-baseline-iters 0 --no-hbm-resident > $OUT/r4_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-200 $OUT/r4_bench_n1_4trials_in_flight.json
