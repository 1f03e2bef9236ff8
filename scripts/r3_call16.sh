# MIOpen solver selection: is it stable across processes on one box, and what do the find modes do to the rate?
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/r3_miopen_selection.txt
census() {  # census <tag> [env...]
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  (cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --cpu-baseline-iters 0 --no-span-timing --no-dry-collective --no-kernel-timing > /tmp/bench_$tag.json 2>/dev/null)
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag ($*)  $(python -c "import json;r=json.load(open('/tmp/bench_$tag.json'));print('under rocprof', r['value'], 'it/s, final objective', r['final_objective'])")" >> $OUT/r3_miopen_selection.txt
  python - "$f" >> $OUT/r3_miopen_selection.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
pick = lambda key: sum(int(r["Calls"]) for r in rows if key in r["Name"])
print(f"   dispatches {tot}  Im2d2Col {pick('Im2d2Col')}  Col2Im {pick('Col2Im')}  Winograd(miopenSp3AsmConv) {pick('miopenSp3AsmConv')}  igemm {pick('igemm')}  ck_grouped_conv {pick('grouped_conv')}  rocBLAS/Tensile Cijk {pick('Cijk')}  naive_conv {pick('naive_conv')}")
PY
}
ls -la ~/.config/miopen ~/.cache/miopen 2>/dev/null | head
census first_process
census second_process
census find_mode_fast MIOPEN_FIND_MODE=2
census find_mode_normal MIOPEN_FIND_MODE=1
ls -la ~/.config/miopen ~/.cache/miopen 2>/dev/null | head -20
cat $OUT/r3_miopen_selection.txt
for m in default 1 2; do
  if [ $m = default ]; then timeout 200 python bench.py --steps 150 --cpu-baseline-iters 0 --no-dry-collective --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('FIND_MODE default', r['value'], r['final_objective'])" >> $OUT/r3_miopen_selection.txt
  else MIOPEN_FIND_MODE=$m timeout 200 python bench.py --steps 150 --cpu-baseline-iters 0 --no-dry-collective --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('FIND_MODE $m', r['value'], r['final_objective'])" >> $OUT/r3_miopen_selection.txt; fi
done
tail -3 $OUT/r3_miopen_selection.txt
