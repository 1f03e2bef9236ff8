set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/r3_gpu_tests_run1.log 2>&1; tail -25 $OUT/r3_gpu_tests_run1.log
timeout 300 python bench.py > $OUT/r3_bench_n1_a.json 2> $OUT/r3_bench_n1_a.err; cut -c1-1500 $OUT/r3_bench_n1_a.json; tail -3 $OUT/r3_bench_n1_a.err
for T in 2 4 8; do timeout 300 python scripts/batched_restarts_probe.py --trials $T >> $OUT/r3_batched_probe.jsonl 2>> $OUT/r3_batched_probe.err; done
cut -c1-400 $OUT/r3_batched_probe.jsonl; tail -5 $OUT/r3_batched_probe.err
timeout 900 python scripts/config_runs.py --full > $OUT/r3_config_runs_same_process.log 2>&1; grep -v Warning $OUT/r3_config_runs_same_process.log | tail -12 | cut -c1-500
