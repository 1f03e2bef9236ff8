set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/r3_gpu_kernel_tests.log 2>&1; tail -3 $OUT/r3_gpu_kernel_tests.log | cut -c1-300
timeout 200 python scripts/replay_threads_probe.py --width 4 > $OUT/r3_replay_threads.jsonl 2> $OUT/r3_replay_threads.err
timeout 200 python scripts/replay_threads_probe.py --width 6 >> $OUT/r3_replay_threads.jsonl 2>> $OUT/r3_replay_threads.err
timeout 200 python scripts/replay_threads_probe.py --width 8 >> $OUT/r3_replay_threads.jsonl 2>> $OUT/r3_replay_threads.err
cat $OUT/r3_replay_threads.jsonl; tail -3 $OUT/r3_replay_threads.err
timeout 400 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -2 $OUT/r3_kernel_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_kernel_bench.json")); print(json.dumps(r.get("kernelD_resnet50_B8")))
for k,v in r.items():
    if isinstance(v, dict) and "after_producer_rewrite" in v: print(k, v["cosine-similarity"], v["after_producer_rewrite"], v["after_1GiB_of_writes"])
PY
