# PMC passes (each in its own rocprofv3 run: --pmc only, no tracing) on the bounded target, plus L2 hit counters in the loop.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
pmc() {  # pmc <seconds> <tag> <counter list> <command...>
  limit=$1; tag=$2; counters=$3; shift 3
  rm -rf /tmp/prof_$tag
  (cd /tmp && timeout $limit rocprofv3 --pmc $counters --output-format csv -d /tmp/prof_$tag -- "$@" > $OUT/${tag}_stdout.log 2> $OUT/${tag}_stderr.log)
  echo "rc=$?" >> $OUT/${tag}_stdout.log
  first=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$first" ]; then python scripts/summarize_prof.py $(dirname $first) $OUT/$tag | head -14; fi
  tail -2 $OUT/${tag}_stdout.log | cut -c1-200
}
T="python $GRAFT_REPO_ROOT/scripts/pmc_target.py"
for size in resnet18 resnet50 bert bn; do
  pmc 150 r3_pmc_fetch_$size FETCH_SIZE $T --size $size
  pmc 150 r3_pmc_write_$size WRITE_SIZE $T --size $size
  pmc 150 r3_pmc_l2_$size "TCC_HIT_sum TCC_MISS_sum" $T --size $size
done
export BREACH_HIP_GRAPH=0
pmc 240 r3_pmc_l2_inloop_resnet50 "TCC_HIT_sum TCC_MISS_sum" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 --its 30
pmc 240 r3_pmc_fetch_inloop_resnet50 FETCH_SIZE python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 --its 30
pmc 240 r3_pmc_l2_inloop_bert "TCC_HIT_sum TCC_MISS_sum" python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5 --its 30
pmc 240 r3_pmc_fetch_inloop_bert FETCH_SIZE python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5 --its 30
ls $OUT | grep pmc_summary
