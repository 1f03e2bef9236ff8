# Round 4, GPU call 16: suite and bench lines with kernel A's size-dependent cache policy.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/r4_gpu_tests_final.log 2>&1; tail -2 $OUT/r4_gpu_tests_final.log | cut -c1-200
timeout 600 python bench.py > $OUT/r4_bench_n1.json 2> $OUT/r4_bench_n1.err; cut -c1-200 $OUT/r4_bench_n1.json
timeout 300 python bench.py --trials-per-gpu 4 --cpu-baseline-iters 0 --no-hbm-resident > $OUT/r4_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-160 $OUT/r4_bench_n1_4trials_in_flight.json
