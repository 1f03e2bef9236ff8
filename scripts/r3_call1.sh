set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 300 python scripts/mode_spread.py > $OUT/r3_mode_spread_convnet.json 2> $OUT/r3_mode_spread_convnet.err; tail -3 $OUT/r3_mode_spread_convnet.err
timeout 900 python -m pytest tests -m gpu -q --durations=15 > $OUT/r3_gpu_tests_run0.log 2>&1; tail -40 $OUT/r3_gpu_tests_run0.log
timeout 330 python scripts/same_process_stall.py --dump-after 100 > $OUT/r3_stall.log 2>&1; tail -60 $OUT/r3_stall.log
