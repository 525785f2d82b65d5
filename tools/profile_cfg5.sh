R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_run.sh cfg5 --mode bf16_1024 --conv-breakdown $R/gpurun_out/prof_cfg5/by_shape.csv > gpurun_out/prof_cfg5.log 2>&1
bash tools/traffic_run.sh cfg5 conv2d_mfma16 --mode bf16_1024 > gpurun_out/traffic_cfg5.log 2>&1
cd $R
python bench.py --mode bf16_1024 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
tail -1 gpurun_out/bench_cfg5.json | cut -c1-300
