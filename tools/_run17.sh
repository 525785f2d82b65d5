cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w17
timeout 900 python -m pytest tests/test_conv16.py tests/test_hip_parity.py -m gpu -x -q -k "channels_last or upfirdn or discriminator or training_iteration" > gpurun_out/w17/tests.log 2>&1; tail -4 gpurun_out/w17/tests.log
for i in 1 2; do python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c75-200; done
