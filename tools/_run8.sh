cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w8
timeout 900 python -m pytest tests/test_conv16.py -m gpu -x -q > gpurun_out/w8/tests.log 2>&1; tail -3 gpurun_out/w8/tests.log
for rep in 1 2; do for v in 0 1; do echo -n "shared=$v "; PG_MODCONV16_SHARED=$v python bench.py --mode bf16_1024 --no-cpu-baseline 2>&1 | tail -1 | cut -c70-190; done; done | tee gpurun_out/w8/ab.log
