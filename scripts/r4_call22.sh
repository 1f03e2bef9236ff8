# Round 4, the very last GPU seconds: kernels E / F with staged loads -- bit-identity against the previous build, replay timing, parity tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 42 python scripts/staged_loads_probe.py --prev build/libbreach_hip_prev.so --pytest > $OUT/r4_staged_loads_probe.jsonl 2> $OUT/r4_staged_loads_probe.err
grep -v "^{\"kernel\"" $OUT/r4_staged_loads_probe.jsonl | cut -c1-600 | tail -12
grep "^{\"kernel\"" $OUT/r4_staged_loads_probe.jsonl | cut -c1-420
tail -3 $OUT/r4_staged_loads_probe.err | cut -c1-300
