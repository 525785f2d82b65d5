cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w10
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "weight_gradient or native_backward" > gpurun_out/w10/tests.log 2>&1; tail -3 gpurun_out/w10/tests.log
for v in 0 1 0 1; do echo -n "native_wgrad=$v "; PG_NATIVE_WGRAD=$v python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c75-200; done | tee gpurun_out/w10/ab.log
