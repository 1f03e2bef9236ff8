set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "eval_batchnorm" > $OUT/r3_evalbn_test.log 2>&1; tail -15 $OUT/r3_evalbn_test.log | cut -c1-300
for mode in hip addcmul; do
  BREACH_HIP_FAST_BN=$mode timeout 200 python bench.py --steps 150 --cpu-baseline-iters 0 --no-dry-collective 2>$OUT/r3_bench_bn_$mode.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$mode', json.dumps(dict(value=r['value'], ms=r['ms_per_step'], mode=r['launch_mode'], final=r['final_objective'], err=r['graph_capture_error'])))"
  tail -2 $OUT/r3_bench_bn_$mode.err | cut -c1-300
done
BREACH_HIP_FAST_BN=hip timeout 200 python bench.py --steps 150 --trials-per-gpu 4 --cpu-baseline-iters 0 --no-dry-collective --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('hip x4', r['value'], r['ms_per_step'])"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/r3_gpu_tests_evalbn.log 2>&1; tail -12 $OUT/r3_gpu_tests_evalbn.log | cut -c1-300
