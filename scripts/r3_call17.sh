set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
R=$OUT/r3_miopen_find_mode.txt
: > $R
run() {  # run <label> [env...]
  label=$1; shift
  echo "== $label" >> $R
  /usr/bin/time -f "   bench wall %e s" -a -o $R env "$@" timeout 300 python bench.py --steps 150 --cpu-baseline-iters 0 --no-dry-collective --no-kernel-timing 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('   bench ResNet-18', r['value'], 'it/s')" >> $R
  /usr/bin/time -f "   config3 wall %e s" -a -o $R env "$@" timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[2\] ResNet-50" | head -1 | cut -c1-180 >> $R
  env "$@" timeout 300 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[2\] ResNet-50" | head -1 | cut -c80-180 >> $R
  ls ~/.config/miopen 2>/dev/null | tr '\n' ' ' >> $R; echo >> $R
}
run "fresh box, MIOPEN_FIND_MODE=2 (FAST)" MIOPEN_FIND_MODE=2
rm -rf ~/.config/miopen
run "find db removed, default find mode" DUMMY=1
cat $R
