set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; tail -2 gpurun_out/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/final_bench.json 2>gpurun_out/final_bench.err; cat gpurun_out/final_bench.json
bash scripts/spmm_pmc.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_q -o q --output-format csv -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_q.log 2>&1
