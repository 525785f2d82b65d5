cd $GRAFT_REPO_ROOT
bash tools/prof_run.sh cfg5 --mode bf16_1024 --conv-breakdown $GRAFT_REPO_ROOT/gpurun_out/prof_cfg5/by_shape.csv > gpurun_out/prof_cfg5.log 2>&1
head -30 gpurun_out/prof_cfg5/kernel_stats_timed.csv | cut -c1-170
