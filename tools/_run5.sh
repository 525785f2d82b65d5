cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w5
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "upfirdn or fir or golden" > gpurun_out/w5/tests.log 2>&1; tail -3 gpurun_out/w5/tests.log
python tools/stream_ops_bw.py 2>&1 | head -6 | tee gpurun_out/w5/bw.log
python bench.py --no-cpu-baseline > gpurun_out/w5/bench.log 2>&1; tail -1 gpurun_out/w5/bench.log | cut -c1-260
