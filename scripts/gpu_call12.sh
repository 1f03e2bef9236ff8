set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 | cut -c1-300
timeout 300 python bench.py > gpurun_out/r2_bench_n1.json 2> gpurun_out/r2_bench_n1.err; cat gpurun_out/r2_bench_n1.json | cut -c1-3000
