set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "layernorm" > $OUT/r3_ln_test.log 2>&1; tail -12 $OUT/r3_ln_test.log | cut -c1-300
R=$OUT/r3_layer_norm_ab.txt
: > $R
for LN in 1 0; do
  echo "== BREACH_HIP_FAST_LN=$LN" >> $R
  BREACH_HIP_FAST_LN=$LN timeout 300 python scripts/config_runs.py --only 5 2>&1 | grep "configs\[4\]" | head -1 | cut -c60-260 >> $R
  BREACH_HIP_FAST_LN=$LN timeout 300 python scripts/config_runs.py --only 5 --its 400 2>&1 | grep "configs\[4\]" | head -1 | cut -c60-260 >> $R
done
cat $R
timeout 600 python -m pytest tests/test_gpu_attack.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "bert or tag" 2>&1 | tail -3 | cut -c1-300
