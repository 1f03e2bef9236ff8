# Round 4, last call (about two GPU-minutes left): the multi-tensor kernel with 16-byte loads -- bit-exactness first, then old vs new.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 55 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_attack.py -x -q -k "multi_tensor or fedavg_multi_step or pearlmutter_objectives" > $OUT/r4_mt_tests.log 2>&1; tail -1 $OUT/r4_mt_tests.log | cut -c1-200
timeout 50 python scripts/mt_kernel_probe.py --prev build/libbreach_mt_prev.so --launches 20 > $OUT/r4_mt_kernel_probe_nt.jsonl 2> $OUT/r4_mt_kernel_probe_nt.err; cut -c1-600 $OUT/r4_mt_kernel_probe_nt.jsonl; tail -2 $OUT/r4_mt_kernel_probe_nt.err | cut -c1-300
