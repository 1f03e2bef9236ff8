set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
rm -rf /tmp/prof_bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-dry-collective > $OUT/r3_bench_under_rocprof.json 2> /dev/null)
first=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_bench | head -14
cp $(dirname $first)/*kernel_stats.csv $OUT/r3_bench_rocprofv3_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r3_bench_kernel_summary.csv")))
its=163
print("calls/it", sum(int(r["calls"]) for r in rows)/its)
rows.sort(key=lambda r:-int(r["calls"]))
for r in rows[:28]: print(f"{int(r['calls'])/its:7.1f}/it avg {float(r['avg_us']):6.2f}us {r['kernel'][:120]}")
PY
