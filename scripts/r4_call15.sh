# Round 4, GPU call 15: kernel A cache policy INSIDE the attack loop: ResNet-18 bench line (events in the eager continuation) and ResNet-50 B = 8 (rocprofv3 kernel trace).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --cpu-baseline-iters 0 --no-dry-collective --no-hbm-resident"
timeout 200 $B --steps 100 > /dev/null 2>&1
for pol in 1,1 2,1 2,2 3,3 1,1 2,1; do
  timeout 200 $B --steps 200 --gm-cache-policy $pol 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('policy $pol: %.1f it/s  fwd %.2f us (frac %.3f)  fin %.2f  bwd %.2f us (frac %.3f)  span-in-replay %.2f us' % (d['value'], k['fwd']['avg_us'], d['roofline']['frac'], k['fin']['avg_us'], k['bwd']['avg_us'], k['bwd']['frac_of_hbm_peak'], d['roofline']['timed_region_span_us']))"
done | tee $OUT/r4_cache_policy_inloop_resnet18.txt
timeout 300 python scripts/config_runs.py --only 3 > /dev/null 2>&1
for pol in 1 2; do
  rm -rf /tmp/prof_c3
  (cd /tmp && GM_CACHE=$pol GM_CACHE_BWD=1 timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c3 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 > $OUT/r4_config3_cache$pol.log 2>&1)
  first=$(ls -S $(find /tmp/prof_c3 -name "*kernel_trace.csv") | head -1)
  python scripts/summarize_prof.py $(dirname $first) $OUT/r4_config3_cache_policy_$pol | grep "gm_\|kernels:"
done
