# (historical: run at commit 32ae57f, where kernel D's fused forward -- BREACH_HIP_BN_FUSED -- still existed; kept as the producer of
#  profiles/r3_kernel_bench_with_fused_bn_forward.json)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/r3_gpu_kernel_tests_fused.log 2>&1; tail -5 $OUT/r3_gpu_kernel_tests_fused.log | cut -c1-300
timeout 400 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -2 $OUT/r3_kernel_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_kernel_bench.json")); print(json.dumps(r.get("kernelD_resnet50_B8")))
for k,v in r.items():
    if isinstance(v, dict) and "after_producer_rewrite" in v: print(k, v["cosine-similarity"], v["after_producer_rewrite"])
PY
BREACH_HIP_BN_FUSED=1 timeout 200 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | cut -c1-300
BREACH_HIP_BN_FUSED=0 timeout 200 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | cut -c1-300
