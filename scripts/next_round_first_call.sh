# First GPU call of the next round: everything that has to be re-measured at the commit round 4 ended on,
# because the multi-tensor kernel and kernels E / F changed after round 4's last complete run.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/next_round_first_call.sh r5'
# Writes gpurun_out/<tag>_*: smoke, the GPU suite (-s: printed evidence), the driver-style bench line, the default bench line, four restarts
# in flight, and the rocprofv3 kernel trace of the bench command with its summary, stats and gap census (copy what is kept to profiles/).
TAG=${1:-r5}
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -1 $OUT/${TAG}_gpu_tests.log | cut -c1-200
timeout 600 $B --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_style.json 2> $OUT/${TAG}_bench_driver_style.err; cut -c1-300 $OUT/${TAG}_bench_driver_style.json
timeout 600 $B > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; cut -c1-300 $OUT/${TAG}_bench_n1.json
timeout 600 $B --trials-per-gpu 4 --cpu-baseline-iters 0 --no-hbm-resident > $OUT/${TAG}_bench_n1_4trials_in_flight.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_n1_4trials_in_flight.json
rm -rf /tmp/prof_bench
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-hbm-resident --no-dry-collective > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_bench_under_rocprof.err)
trace=$(ls -S $(find /tmp/prof_bench -name "*kernel_trace.csv") | head -1)
if [ -n "$trace" ]; then
  python scripts/summarize_prof.py $(dirname $trace) $OUT/${TAG}_bench | head -16
  stats=$(ls -S $(find /tmp/prof_bench -name "*kernel_stats.csv") | head -1); [ -n "$stats" ] && cp "$stats" $OUT/${TAG}_bench_rocprofv3_kernel_stats.csv
  python scripts/gap_census.py $trace $OUT/${TAG}_1trial_gap_census --iters 60 --skip-tail 45 --label "1 trial, staged loads in kernels E / F" | head -28
fi
