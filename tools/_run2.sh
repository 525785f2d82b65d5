cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wino or conv2d or modulated or spade or golden" > gpurun_out/w2/tests.log 2>&1; tail -3 gpurun_out/w2/tests.log
python bench.py --no-cpu-baseline --conv-breakdown gpurun_out/w2/bd.csv > gpurun_out/w2/bench.log 2>&1; tail -1 gpurun_out/w2/bench.log | cut -c1-900
