cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -- python3 $R/bench.py --no-cpu-baseline --steps 7 --warmup 2 > $O/bench_under_rocprof.log 2>&1
f=$(find /tmp/prof_out -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
tail -1 $O/bench_under_rocprof.log | cut -c1-400
head -40 $O/kernel_stats.csv
