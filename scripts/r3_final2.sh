set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
for k in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3_gpu_tests_final_$k.log 2>&1; tail -2 $OUT/r3_gpu_tests_final_$k.log | cut -c1-200
done
grep "call " $OUT/r3_gpu_tests_final_1.log | head
