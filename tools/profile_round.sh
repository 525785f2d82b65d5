# GPU box, one gpurun call: everything profiles/rNN_* is made from (tools/collect_profiles.py turns it into the committed files).
#   bash tools/profile_round.sh
R=$GRAFT_REPO_ROOT
cd $R
# build the plugins in a plain, unprofiled process first: under rocprofv3 the loader's hipcc children would inherit the profiler's preload
python3 tools/ensure_built.py > gpurun_out/build.log 2>&1
bash tools/prof_run.sh cfg2 --conv-breakdown $R/gpurun_out/prof_cfg2/by_shape.csv > gpurun_out/prof_cfg2.log 2>&1
bash tools/prof_run.sh cfg5 --mode bf16_1024 --conv-breakdown $R/gpurun_out/prof_cfg5/by_shape.csv > gpurun_out/prof_cfg5.log 2>&1
bash tools/prof_run.sh train --mode train --steps 3 --warmup 1 > gpurun_out/prof_train.log 2>&1
bash tools/traffic_run.sh cfg2 conv2d_wino4 > gpurun_out/traffic_cfg2.log 2>&1
bash tools/traffic_run.sh cfg5 pgconv16 --mode bf16_1024 > gpurun_out/traffic_cfg5.log 2>&1
bash tools/traffic_run.sh train 'wgrad|split3_bf16' --mode train > gpurun_out/traffic_train.log 2>&1
rm -f gpurun_out/pmc/summary.txt
bash tools/pmc_run.sh > gpurun_out/pmc_wino.log 2>&1
bash tools/pmc16_run.sh 4 64 64 512 > gpurun_out/pmc16.log 2>&1
cd $R
python bench.py > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --mode bf16_1024 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err
python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err
tail -1 gpurun_out/bench_cfg2.json | cut -c1-400; tail -1 gpurun_out/bench_cfg5.json | cut -c1-400; tail -1 gpurun_out/bench_train.json | cut -c1-400
