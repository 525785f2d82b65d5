cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w9
timeout 900 python -m pytest tests/test_conv16.py -m gpu -x -q -k "graphed or stack" > gpurun_out/w9/tests.log 2>&1; tail -3 gpurun_out/w9/tests.log
python bench.py --mode bf16_1024 --no-cpu-baseline > gpurun_out/w9/graph.log 2>&1; tail -1 gpurun_out/w9/graph.log | cut -c1-300; tail -1 gpurun_out/w9/graph.log | grep -o '"roofline".*' | cut -c1-700
