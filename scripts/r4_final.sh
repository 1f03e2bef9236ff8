# Round 4, final validation at HEAD: smoke, the GPU suite twice (flake check), the driver-style bench line.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python __graft_entry__.py smoke > $OUT/r4_smoke.log 2>&1; tail -1 $OUT/r4_smoke.log | cut -c1-200
for k in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r4_gpu_tests_final_$k.log 2>&1; tail -1 $OUT/r4_gpu_tests_final_$k.log | cut -c1-200
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r4_bench_driver_style.json 2> $OUT/r4_bench_driver_style.err; cut -c1-300 $OUT/r4_bench_driver_style.json
timeout 600 python bench.py > $OUT/r4_bench_n1.json 2> $OUT/r4_bench_n1.err; cut -c1-300 $OUT/r4_bench_n1.json
