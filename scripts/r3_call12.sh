set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/r3_gpu_kernel_tests.log 2>&1; tail -4 $OUT/r3_gpu_kernel_tests.log | cut -c1-300
timeout 500 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -2 $OUT/r3_kernel_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_kernel_bench.json"))
for k,v in r.items():
    if isinstance(v, dict) and "fwd_us_by_pipeline_mode" in v: print(k, v["cosine-similarity"]["fwd_us"], v["fwd_us_by_pipeline_mode"])
PY
