# Round 4, GPU call 8: CU-mask probe (does a trial in flight run faster on its own slice of the chip?), the adjusted 24k end-of-run test.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python scripts/cu_mask_probe.py > $OUT/r4_cu_mask_probe.jsonl 2> $OUT/r4_cu_mask_probe.err; cut -c1-300 $OUT/r4_cu_mask_probe.jsonl; tail -3 $OUT/r4_cu_mask_probe.err | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "24k_end_of_run" > $OUT/r4_gpu_tests_24k_end.log 2>&1; tail -14 $OUT/r4_gpu_tests_24k_end.log | cut -c1-260
