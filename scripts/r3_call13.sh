set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/r3_two_processes_one_gpu.jsonl
for cfg in "2 2" "2 4" "4 1" "4 2" "3 3"; do set -- $cfg
  timeout 300 python bench.py --gpus $1 --trials-per-gpu $2 --steps 100 --cpu-baseline-iters 0 --no-kernel-timing 2>/dev/null | grep '^{"metric' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(json.dumps(dict(processes=r['n_gpus'], trials_in_flight_per_process=r['config']['trials_in_flight_per_gpu'], value=r['value'], ms_per_step=r['ms_per_step'], oversubscribed=r['oversubscribed'])))" >> $OUT/r3_two_processes_one_gpu.jsonl
done
cat $OUT/r3_two_processes_one_gpu.jsonl
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2 | cut -c1-200
