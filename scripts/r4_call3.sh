# Round 4, GPU call 3: kernel E epilogue (BN + residual + ReLU) tests and A/B on the bench line, pipe-aware side streams, new bench legs.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 300 python __graft_entry__.py smoke > $OUT/r4_smoke.log 2>&1; tail -1 $OUT/r4_smoke.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "epilogue or fused or tap or deepinversion or eval_batchnorm" > $OUT/r4_gpu_tests_call3_kernels.log 2>&1; tail -12 $OUT/r4_gpu_tests_call3_kernels.log | cut -c1-250
timeout 200 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 100 > /dev/null 2>&1
for mode in 1 0 1 0; do
  BREACH_HIP_FUSE_BN_RELU=$mode timeout 200 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 200 > $OUT/r4_ab_fuse_bn_relu_$mode.json 2>$OUT/r4_ab_fuse_err.log; cut -c1-140 $OUT/r4_ab_fuse_bn_relu_$mode.json; tail -2 $OUT/r4_ab_fuse_err.log | cut -c1-300
done
for mode in 1 0; do
  BREACH_HIP_FUSE_BN_RELU=$mode timeout 200 $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 200 --trials-per-gpu 4 > $OUT/r4_ab_fuse_bn_relu_4trials_$mode.json 2>/dev/null; cut -c1-140 $OUT/r4_ab_fuse_bn_relu_4trials_$mode.json
done
timeout 1800 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/r4_gpu_tests_call3.log 2>&1; tail -25 $OUT/r4_gpu_tests_call3.log | cut -c1-250
rm -rf /tmp/prof_fused
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_fused -- $B --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident --steps 100 --warmup 20 --no-kernel-timing --no-span-timing > $OUT/r4_fused_stdout.log 2> $OUT/r4_fused_stderr.log)
first=$(find /tmp/prof_fused -name "*kernel_trace.csv" | head -1)
[ -n "$first" ] && python scripts/gap_census.py $first $OUT/r4_1trial_fused_gap_census --iters 60 --label "1 trial, BN + residual + ReLU in kernel E" | head -30
timeout 400 python scripts/op_attribution.py > $OUT/r4_op_attribution_fused.json 2> $OUT/r4_op_attribution_fused.txt; head -30 $OUT/r4_op_attribution_fused.txt
timeout 600 $B > $OUT/r4_bench_n1.json 2> $OUT/r4_bench_n1.err; cut -c1-400 $OUT/r4_bench_n1.json; tail -3 $OUT/r4_bench_n1.err
timeout 400 python scripts/cpu_thread_sweep.py > $OUT/r4_cpu_thread_sweep.json 2> $OUT/r4_cpu_thread_sweep.err; cat $OUT/r4_cpu_thread_sweep.err | tail -8
