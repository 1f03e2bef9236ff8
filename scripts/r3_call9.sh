set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/r3_gpu_tests_run3.log 2>&1; tail -16 $OUT/r3_gpu_tests_run3.log | cut -c1-300
timeout 400 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -2 $OUT/r3_kernel_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_kernel_bench.json")); print(json.dumps(r.get("kernelD_resnet50_B8")))
PY
timeout 200 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | cut -c1-300
rm -rf /tmp/prof_c3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > $OUT/r3_config3_prof.log 2>&1)
first=$(find /tmp/prof_c3 -name "*kernel_trace.csv" | head -1)
[ -n "$first" ] && python scripts/summarize_prof.py $(dirname $first) $OUT/r3_config3_resnet50_seethrough | head -14
