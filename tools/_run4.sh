cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_conv16.py -m gpu -x -q -k "backward or weight_gradient or gradfix or grad or stack" > gpurun_out/w4/tests.log 2>&1; tail -3 gpurun_out/w4/tests.log
python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/w4/train_native.log 2>&1; tail -1 gpurun_out/w4/train_native.log | cut -c1-300
PG_NATIVE_WGRAD=0 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/w4/train_aten_w.log 2>&1; tail -1 gpurun_out/w4/train_aten_w.log | cut -c1-300
python bench.py --mode bf16_1024 --no-cpu-baseline > gpurun_out/w4/cfg5.log 2>&1; tail -1 gpurun_out/w4/cfg5.log | cut -c1-300
