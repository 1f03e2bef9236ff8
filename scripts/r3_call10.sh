# Same-box A/B of kernel D's forward stage INSIDE the attack loop (ResNet-50, B = 8, see-through + DeepInversion) under
# rocprofv3 --kernel-trace: loads in flight 4 vs 8, uncapped vs 2048-workgroup grid, finalize block 256 / 512 / 1024.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/r3_bn_inloop_ab.txt
ab() {  # ab <tag> <env assignments...>
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  (cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 --its 80 > /dev/null 2>&1)
  first=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  if [ -n "$first" ]; then
    python scripts/summarize_prof.py $(dirname $first) /tmp/sum_$tag > /dev/null
    echo "== $tag ($*)" >> $OUT/r3_bn_inloop_ab.txt
    grep "bn_\|gm_fwd\|gm_bwd" /tmp/sum_${tag}_kernel_summary.txt >> $OUT/r3_bn_inloop_ab.txt
  fi
}
ab depth8_uncapped_fin1024 BN_DEPTH=8
ab depth4_uncapped_fin1024 BN_DEPTH=4
ab depth8_cap2048_fin1024 BN_DEPTH=8 BN_GRID_CAP=2048
ab depth8_uncapped_fin256 BN_DEPTH=8 BN_FIN_BLOCK=256
ab depth8_uncapped_fin512 BN_DEPTH=8 BN_FIN_BLOCK=512
ab depth4_cap4096_fin1024 BN_DEPTH=4 BN_GRID_CAP=4096
cat $OUT/r3_bn_inloop_ab.txt | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -x -q -k "fedavg or worker" 2>&1 | tail -3 | cut -c1-300
