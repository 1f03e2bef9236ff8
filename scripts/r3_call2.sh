set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
L=$OUT/r3_stall_bisect.jsonl
: > $L
S="timeout 150 python scripts/stall_bisect.py"
$S --first none --repeat 2 >> $L 2>$OUT/stall_err.log
$S --first resnet50 >> $L 2>>$OUT/stall_err.log
$S --first convnet >> $L 2>>$OUT/stall_err.log
$S --first resnet18 >> $L 2>>$OUT/stall_err.log
$S --first resnet50-nodi >> $L 2>>$OUT/stall_err.log
$S --first resnet50-nograph >> $L 2>>$OUT/stall_err.log
$S --first resnet50 --empty-cache >> $L 2>>$OUT/stall_err.log
GPU_MAX_HW_QUEUES=8 $S --first resnet50 >> $L 2>>$OUT/stall_err.log
$S --first resnet50 --width 1 >> $L 2>>$OUT/stall_err.log
cat $L | cut -c1-600
tail -5 $OUT/stall_err.log
