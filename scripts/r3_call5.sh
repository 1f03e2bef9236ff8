set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/r3_gpu_tests_run2.log 2>&1; tail -25 $OUT/r3_gpu_tests_run2.log | cut -c1-300
timeout 300 python scripts/kernel_bench.py > $OUT/r3_kernel_bench.json 2> $OUT/r3_kernel_bench.err; tail -2 $OUT/r3_kernel_bench.err; python - <<'PY'
import json
r=json.load(open("gpurun_out/r3_kernel_bench.json")); print(json.dumps(r.get("kernelD_resnet50_B8")))
PY
: > $OUT/r3_inflight_width.jsonl
for W in 4 6 8; do timeout 200 python bench.py --trials-per-gpu $W --steps 150 --cpu-baseline-iters 0 --no-kernel-timing --no-dry-collective 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps(dict(trials_in_flight=r['config']['trials_in_flight_per_gpu'], value=r['value'], ms_per_step=r['ms_per_step'])))" >> $OUT/r3_inflight_width.jsonl; done
GPU_MAX_HW_QUEUES=16 timeout 200 python bench.py --trials-per-gpu 8 --steps 150 --cpu-baseline-iters 0 --no-kernel-timing --no-dry-collective 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps(dict(hwq=16, trials_in_flight=r['config']['trials_in_flight_per_gpu'], value=r['value'], ms_per_step=r['ms_per_step'])))" >> $OUT/r3_inflight_width.jsonl
cat $OUT/r3_inflight_width.jsonl
for G in "4 2" "4 4" "8 2" "8 4"; do set -- $G; timeout 300 python scripts/batched_restarts_probe.py --trials $1 --groups $2 >> $OUT/r3_batched_probe_groups.jsonl 2>> $OUT/r3_batched_probe.err; done
cat $OUT/r3_batched_probe_groups.jsonl
for V in separate batched; do
  rm -rf /tmp/prof_$V
  (cd /tmp && timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -- python $GRAFT_REPO_ROOT/scripts/batched_restarts_probe.py --trials 4 --steps 20 --only $V > $OUT/r3_batched_prof_$V.log 2>&1)
  first=$(find /tmp/prof_$V -name "*kernel_stats.csv" | head -1)
  [ -n "$first" ] && cp $first $OUT/r3_batched_prof_${V}_kernel_stats.csv && head -12 $first | cut -c1-200
done
