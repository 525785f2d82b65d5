cd $GRAFT_REPO_ROOT
bash tools/prof_run.sh train --mode train --steps 3 --warmup 3 > gpurun_out/prof_train.log 2>&1
head -45 gpurun_out/prof_train/kernel_stats_timed.csv | cut -c1-160
