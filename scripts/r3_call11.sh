# Before / after of "taking the accumulate back" (DeepInversion backward), same box: ResNet-50 B=8 see-through, rocprofv3
# kernel trace of 80 iterations each way, then the plain rate of 200 iterations each way.  Then bench / profiles of the round.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
: > $OUT/r3_bn_backward_ab.txt
for TAPS in 1 0; do
  tag=taps$TAPS
  rm -rf /tmp/prof_$tag
  (cd /tmp && BREACH_HIP_BN_TAPS=$TAPS timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 3 --its 80 > /dev/null 2>&1)
  first=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
  python scripts/summarize_prof.py $(dirname $first) /tmp/sum_$tag > /dev/null
  echo "== BREACH_HIP_BN_TAPS=$TAPS" >> $OUT/r3_bn_backward_ab.txt
  head -1 /tmp/sum_${tag}_kernel_summary.txt >> $OUT/r3_bn_backward_ab.txt
  grep "bn_" /tmp/sum_${tag}_kernel_summary.txt >> $OUT/r3_bn_backward_ab.txt
  grep "CUDAFunctor_add<float>" /tmp/sum_${tag}_kernel_summary.csv | cut -c1-160 >> $OUT/r3_bn_backward_ab.txt
  BREACH_HIP_BN_TAPS=$TAPS timeout 200 python scripts/config_runs.py --only 3 2>&1 | grep "configs\[" | cut -c1-200 >> $OUT/r3_bn_backward_ab.txt
done
cat $OUT/r3_bn_backward_ab.txt
timeout 300 python bench.py > $OUT/r3_bench_n1.json 2> $OUT/r3_bench_n1.err; cut -c1-600 $OUT/r3_bench_n1.json
timeout 300 python bench.py --gpus 2 --steps 50 --cpu-baseline-iters 0 > $OUT/r3_bench_2ranks_one_gpu.json 2> $OUT/r3_bench_2ranks_one_gpu.err; cut -c1-400 $OUT/r3_bench_2ranks_one_gpu.json; tail -2 $OUT/r3_bench_2ranks_one_gpu.err | cut -c1-300
timeout 200 python bench.py --trials-per-gpu 4 --cpu-baseline-iters 0 --no-dry-collective > $OUT/r3_bench_n1_4trials_in_flight.json 2>/dev/null; cut -c1-300 $OUT/r3_bench_n1_4trials_in_flight.json
rm -rf /tmp/prof_bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --cpu-baseline-iters 0 --no-span-timing --no-dry-collective > $OUT/r3_bench_under_rocprof.json 2> /dev/null)
first=$(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_bench | head -12
cp $(dirname $first)/*kernel_stats.csv $OUT/r3_bench_rocprofv3_kernel_stats.csv
rm -rf /tmp/prof_c5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/scripts/config_runs.py --only 5 > /dev/null 2>&1)
first=$(find /tmp/prof_c5 -name "*kernel_trace.csv" | head -1)
python scripts/summarize_prof.py $(dirname $first) $OUT/r3_config5_bert_tag | head -8
