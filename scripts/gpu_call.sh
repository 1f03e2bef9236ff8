# One GPU call = a list of named stages (round 5 replaces the per-call r3_/r4_ scripts by this one file):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/gpu_call.sh r5a kernels ceiling mt stepprior'
# First argument = tag (prefix of everything written under gpurun_out/), then stages in order.  Every stage runs under its own
# `timeout`: a hung pass must not eat the GPU budget.  Summaries that are kept are copied to profiles/ by hand.
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py"
PREV=$GRAFT_REPO_ROOT/build/libbreach_hip_prev.so

prof() {  # prof <seconds> <name> <pmc-counter or ""> <command...>: rocprofv3 kernel trace (+stats) or one PMC pass, summarised
  limit=$1; name=${TAG}_$2; counter=$3; shift 3
  rm -rf /tmp/prof_$name
  if [ -z "$counter" ]; then
    (cd /tmp && timeout $limit rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $OUT/${name}_stdout.log 2> $OUT/${name}_stderr.log)
  else
    (cd /tmp && timeout $limit rocprofv3 --pmc $counter --output-format csv -d /tmp/prof_$name -- "$@" > $OUT/${name}_stdout.log 2> $OUT/${name}_stderr.log)
  fi
  first=$(find /tmp/prof_$name -name "*.csv" | head -1)
  if [ -n "$first" ]; then
    dir=$(dirname $first)
    python scripts/summarize_prof.py $dir $OUT/$name $counter | head -14
    cp $dir/*kernel_stats.csv $OUT/${name}_rocprofv3_kernel_stats.csv 2>/dev/null
  fi
  tail -1 $OUT/${name}_stdout.log | cut -c1-300
}

example()  { timeout 300 python examples/minimal_example.py > $OUT/${TAG}_minimal_example.log 2>&1; tail -2 $OUT/${TAG}_minimal_example.log | cut -c1-300; }
smoke()    { timeout 300 python __graft_entry__.py smoke > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log | cut -c1-200; }
kernels()  { timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q > $OUT/${TAG}_gpu_tests_kernels.log 2>&1; tail -3 $OUT/${TAG}_gpu_tests_kernels.log | cut -c1-250; }
newtests() { timeout 900 python -m pytest tests -m gpu -x -q -s -k "$NEWTESTS" > $OUT/${TAG}_gpu_tests_new.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests_new.log | cut -c1-300; }
suite()    { timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -1 $OUT/${TAG}_gpu_tests.log | cut -c1-200; }
ceiling()  { timeout 300 python scripts/read_ceiling_probe.py --launches 30 > $OUT/${TAG}_read_ceiling_probe.jsonl 2> $OUT/${TAG}_read_ceiling_probe.err; grep -c . $OUT/${TAG}_read_ceiling_probe.jsonl; grep "kernel A" $OUT/${TAG}_read_ceiling_probe.jsonl | cut -c1-400; tail -2 $OUT/${TAG}_read_ceiling_probe.err | cut -c1-300; }
mt()       { timeout 200 python scripts/mt_kernel_probe.py --launches 30 > $OUT/${TAG}_mt_kernel_probe.jsonl 2> $OUT/${TAG}_mt_kernel_probe.err; cut -c1-700 $OUT/${TAG}_mt_kernel_probe.jsonl; tail -2 $OUT/${TAG}_mt_kernel_probe.err | cut -c1-300; }
stepprior() {
  timeout 120 python scripts/step_prior_probe.py > $OUT/${TAG}_step_prior_probe.jsonl 2> $OUT/${TAG}_step_prior_probe.err
  BREACH_HIP_LIB=$PREV timeout 120 python scripts/step_prior_probe.py >> $OUT/${TAG}_step_prior_probe.jsonl 2>> $OUT/${TAG}_step_prior_probe.err
  cut -c1-330 $OUT/${TAG}_step_prior_probe.jsonl; tail -2 $OUT/${TAG}_step_prior_probe.err | cut -c1-300
}
bench_driver() { timeout 600 $B --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_style.json 2> $OUT/${TAG}_bench_driver_style.err; cut -c1-600 $OUT/${TAG}_bench_driver_style.json; tail -2 $OUT/${TAG}_bench_driver_style.err | cut -c1-300; }
bench_n1()     { timeout 600 $B > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; cut -c1-300 $OUT/${TAG}_bench_n1.json; }
bench_4()      { timeout 600 $B --trials-per-gpu 4 --cpu-baseline-iters 0 --no-hbm-resident > $OUT/${TAG}_bench_n1_4trials_in_flight.json 2> /dev/null; cut -c1-200 $OUT/${TAG}_bench_n1_4trials_in_flight.json; }
bench_8ranks() { timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 --restarts32-iters ${POOL_ITERS:-100} > $OUT/${TAG}_bench_8ranks_one_gpu.json 2> $OUT/${TAG}_bench_8ranks_one_gpu.err; tail -1 $OUT/${TAG}_bench_8ranks_one_gpu.json | cut -c1-900; tail -3 $OUT/${TAG}_bench_8ranks_one_gpu.err | cut -c1-300; }
trace_bench()  {
  prof 400 bench "" $B --steps 100 --warmup 20 --cpu-baseline-iters 0 --gpu-torch-baseline-iters 0 --no-parity --no-span-timing --no-hbm-resident --no-dry-collective --restarts32-iters 0
  trace=$(ls -S $(find /tmp/prof_${TAG}_bench -name "*kernel_trace.csv") | head -1)
  [ -n "$trace" ] && python scripts/gap_census.py $trace $OUT/${TAG}_1trial_gap_census --iters 60 --skip-tail 45 --label "1 trial, round 6 HEAD" | head -20
}
trace_bench_eager() {  # the launch mode bench.py's event-timed roofline iterations run in (a replayed graph cannot carry event pairs)
  prof 400 bench_eager "" $B --steps 100 --warmup 20 --no-graph --cpu-baseline-iters 0 --gpu-torch-baseline-iters 0 --no-parity --no-span-timing --no-hbm-resident --no-dry-collective --restarts32-iters 0
}
trace5()   { prof 400 config5_bert_tag "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5; }
trace3()   { prof 400 config3_resnet50_seethrough "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3; }
trace_fedavg() {
  prof 400 fedavg_resnet18 "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only fedavg --its 100
  FEDAVG_MODEL=resnet50 FEDAVG_LR=0.0005 prof 400 fedavg_resnet50 "" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only fedavg --its 60
}
pmc()      {  # pmc <size>: FETCH_SIZE and WRITE_SIZE in separate passes over scripts/pmc_target.py --size <size>
  prof 200 pmc_fetch_$1 FETCH_SIZE python $GRAFT_REPO_ROOT/scripts/pmc_target.py --size $1
  prof 200 pmc_write_$1 WRITE_SIZE python $GRAFT_REPO_ROOT/scripts/pmc_target.py --size $1
}
pmc_bert()     { pmc bert; }
pmc_resnet50() { pmc resnet50; }
pmc_resnet18() { pmc resnet18; }
pmc_mt()       { pmc mt_resnet50; pmc mt_bert; }
cfg24k()   { timeout 900 python scripts/config_runs.py --only 24k --starts 8 > $OUT/${TAG}_config1_24k_8starts.log 2>&1; tail -1 $OUT/${TAG}_config1_24k_8starts.log > $OUT/${TAG}_config1_24k_8starts.json; grep "hip .* reference" $OUT/${TAG}_config1_24k_8starts.log | cut -c1-250; }
configs()  { timeout 1500 python scripts/config_runs.py --full --its 1000 > $OUT/${TAG}_config_runs_same_process.log 2>&1; tail -12 $OUT/${TAG}_config_runs_same_process.log | cut -c1-400; }
control()  { timeout 900 python tests/control_same_gpu_torch.py --out $OUT/${TAG}_control_same_gpu_torch.json > $OUT/${TAG}_control_same_gpu_torch.log 2>&1; tail -14 $OUT/${TAG}_control_same_gpu_torch.log | cut -c1-260; }
torch24()  { timeout 900 python tests/control_same_gpu_torch.py --torch-only --first-start 8 --starts 24 --processes 3 --out $OUT/${TAG}_torch_on_gpu_24_more_starts.json > $OUT/${TAG}_torch_on_gpu_24_more_starts.log 2>&1; tail -10 $OUT/${TAG}_torch_on_gpu_24_more_starts.log | cut -c1-200; }
tagtwin()  { timeout 600 python scripts/tag_twin_probe.py > $OUT/${TAG}_tag_twin_probe.jsonl 2> $OUT/${TAG}_tag_twin_probe.err; cut -c1-600 $OUT/${TAG}_tag_twin_probe.jsonl; tail -2 $OUT/${TAG}_tag_twin_probe.err | cut -c1-200; }
hip64()    { timeout 600 python tests/control_same_gpu_torch.py --hip-only --starts 64 --out $OUT/${TAG}_hip_64starts_1000its.json > $OUT/${TAG}_hip_64starts_1000its.log 2>&1; tail -10 $OUT/${TAG}_hip_64starts_1000its.log | cut -c1-200; }
stepprior_trace() {
  prof 200 step_prior_probe "" python $GRAFT_REPO_ROOT/scripts/step_prior_probe.py --launches 20
  BREACH_HIP_LIB=$PREV prof 200 step_prior_probe_prev "" python $GRAFT_REPO_ROOT/scripts/step_prior_probe.py --launches 20
}
mt_trace() { prof 200 mt_kernel_probe "" python $GRAFT_REPO_ROOT/scripts/mt_kernel_probe.py --launches 20; }

# ---- round 6 ----
suite_all() { timeout 3000 python -m pytest tests -m gpu -q -s > $OUT/${TAG}_gpu_tests.log 2>&1; tail -1 $OUT/${TAG}_gpu_tests.log | cut -c1-200; grep "^FAILED" $OUT/${TAG}_gpu_tests.log | cut -c1-200; }
affine()   {  # kernel E layer by layer at B = 8: rocprofv3 kernel trace of scripts/affine_layer_probe.py joined with its manifest
  rm -rf /tmp/prof_${TAG}_affine
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_${TAG}_affine -- python $GRAFT_REPO_ROOT/scripts/affine_layer_probe.py --manifest $OUT/${TAG}_affine_layer_manifest.json > $OUT/${TAG}_affine_stdout.log 2> $OUT/${TAG}_affine_stderr.log)
  trace=$(ls -S $(find /tmp/prof_${TAG}_affine -name "*kernel_trace.csv") | head -1)
  [ -n "$trace" ] && python scripts/affine_layer_probe.py --join $trace --manifest $OUT/${TAG}_affine_layer_manifest.json --out $OUT/${TAG}_affine_layers 2>&1 | cut -c1-900
  tail -2 $OUT/${TAG}_affine_stderr.log | cut -c1-300
}
affine_pmc() {  # HBM traffic of kernel E per layer: FETCH_SIZE and WRITE_SIZE in separate counter-only passes over the same probe
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof_${TAG}_affine_$c
    (cd /tmp && timeout 400 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_${TAG}_affine_$c -- python $GRAFT_REPO_ROOT/scripts/affine_layer_probe.py --reps 4 --manifest $OUT/${TAG}_affine_pmc_manifest.json > $OUT/${TAG}_affine_pmc_$c.log 2>&1)
  done
  f=$(find /tmp/prof_${TAG}_affine_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find /tmp/prof_${TAG}_affine_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && [ -n "$w" ] && python scripts/affine_layer_probe.py --join-pmc $f $w --manifest $OUT/${TAG}_affine_pmc_manifest.json --out $OUT/${TAG}_affine_layers 2>&1 | cut -c1-900
}
pool8()    { timeout 900 $B --restarts32-pool 8 --restarts32-iters ${POOL_ITERS:-100} > $OUT/${TAG}_restarts32_pool8_one_gpu.json 2> $OUT/${TAG}_restarts32_pool8_one_gpu.err; tail -1 $OUT/${TAG}_restarts32_pool8_one_gpu.json | cut -c1-1500; tail -3 $OUT/${TAG}_restarts32_pool8_one_gpu.err | cut -c1-300; }
pool2()    { timeout 900 $B --restarts32-pool 2 --restarts32-iters ${POOL_ITERS:-1000} > $OUT/${TAG}_restarts32_pool2_one_gpu.json 2> $OUT/${TAG}_restarts32_pool2_one_gpu.err; tail -1 $OUT/${TAG}_restarts32_pool2_one_gpu.json | cut -c1-1500; tail -3 $OUT/${TAG}_restarts32_pool2_one_gpu.err | cut -c1-300; }
tailprobe() { timeout 200 python scripts/tail_probe.py > $OUT/${TAG}_tail_probe.jsonl 2> $OUT/${TAG}_tail_probe.err; cat $OUT/${TAG}_tail_probe.jsonl; tail -2 $OUT/${TAG}_tail_probe.err | cut -c1-300; }
batched()  {  # trial batching decided with a number: vmap(grad) over K restarts in one victim pass vs 4 in flight (553 it/s)
  for k in 4 8 16; do
    timeout 300 python scripts/batched_restarts_probe.py --trials $k --steps 40 --only batched >> $OUT/${TAG}_batched_restarts_probe.jsonl 2>> $OUT/${TAG}_batched_restarts_probe.err
  done
  timeout 300 python scripts/batched_restarts_probe.py --trials 8 --steps 40 --only batched --groups 2 >> $OUT/${TAG}_batched_restarts_probe.jsonl 2>> $OUT/${TAG}_batched_restarts_probe.err
  cut -c1-400 $OUT/${TAG}_batched_restarts_probe.jsonl; tail -2 $OUT/${TAG}_batched_restarts_probe.err | cut -c1-300
}

set -x
for stage in "$@"; do $stage; done
