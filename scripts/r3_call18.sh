set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "eval_batchnorm" > $OUT/r3_evalbn_test.log 2>&1; tail -6 $OUT/r3_evalbn_test.log | cut -c1-300
R=$OUT/r3_eval_bn_modes.txt
: > $R
for mode in hip addcmul; do
  echo "== BREACH_HIP_FAST_BN=$mode" >> $R
  BREACH_HIP_FAST_BN=$mode timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[2\] ResNet-50" | head -1 | cut -c80-200 >> $R
  BREACH_HIP_FAST_BN=$mode timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[2\] ResNet-50" | head -1 | cut -c80-200 >> $R
  BREACH_HIP_FAST_BN=$mode timeout 200 python bench.py --steps 150 --cpu-baseline-iters 0 --no-dry-collective --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('   bench ResNet-18', r['value'], 'it/s')" >> $R
  BREACH_HIP_FAST_BN=$mode timeout 300 python scripts/config_runs.py --only 1 2>&1 | grep "configs\[0\]" | head -1 | cut -c50-160 >> $R
done
cat $R
