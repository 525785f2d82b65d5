cd $GRAFT_REPO_ROOT
export PG_NATIVE_WGRAD=1
bash tools/prof_run.sh trainw --mode train --steps 3 --warmup 3 > gpurun_out/prof_trainw.log 2>&1
grep -i "wgrad\|igemm_wrw\|batched_transpose\|^#" gpurun_out/prof_trainw/kernel_stats_timed.csv | cut -c1-200
