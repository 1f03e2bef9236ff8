# Round 4, GPU call 10: kernel E with the folded input-gradient accumulation: kernel tests, A/B against the previous commit's library is not
# possible in one checkout, so the bench line + dispatch count are compared with call 7's (235.4 it/s, 590 dispatches); then the whole suite.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x > $OUT/r4_gpu_tests_call10_kernels.log 2>&1; tail -4 $OUT/r4_gpu_tests_call10_kernels.log | cut -c1-250
timeout 300 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 100 > /dev/null 2>&1
for k in 1 2; do timeout 300 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 200 2>/dev/null | cut -c1-140; done
timeout 300 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 200 --trials-per-gpu 4 2>/dev/null | cut -c1-140
timeout 400 python scripts/op_attribution.py > $OUT/r4_op_attribution_folded.json 2> $OUT/r4_op_attribution_folded.txt; grep "launches per iteration\|add" $OUT/r4_op_attribution_folded.txt | head -12
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/r4_gpu_tests_call10.log 2>&1; tail -3 $OUT/r4_gpu_tests_call10.log | cut -c1-200
