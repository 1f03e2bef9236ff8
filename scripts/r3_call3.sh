set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
L=$OUT/r3_stall_bisect2.jsonl
: > $L
S="timeout 150 python scripts/stall_bisect.py"
$S --first convnet --reset-capture-stream >> $L 2>$OUT/stall_err.log
$S --first convnet --burn-streams 1 >> $L 2>>$OUT/stall_err.log
$S --first convnet --burn-streams 4 >> $L 2>>$OUT/stall_err.log
$S --first none --burn-streams 1 >> $L 2>>$OUT/stall_err.log
BREACH_HIP_GROUP_MAIN_STREAM=0 $S --first convnet >> $L 2>>$OUT/stall_err.log
BREACH_HIP_GROUP_MAIN_STREAM=0 $S --first none >> $L 2>>$OUT/stall_err.log
cat $L | cut -c1-900
for v in convnet none; do
  rm -rf /tmp/prof_$v
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$v -- python $GRAFT_REPO_ROOT/scripts/stall_bisect.py --first $v --its 50 > $OUT/r3_trace_$v.log 2>&1)
  python scripts/trace_queues.py /tmp/prof_$v 0.4 | tee $OUT/r3_trace_queues_$v.json | cut -c1-1500
done
tail -3 $OUT/stall_err.log
