# usage (on the GPU box): bash tools/prof_run.sh <tag> <bench.py args...>
#   -> gpurun_out/prof_<tag>/kernel_stats.csv (whole process) and kernel_stats_timed.csv (timed region only, tools/window_stats.py)
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/ensure_built.py || exit 1     # plugins are built in a plain process, never under the profiler
R=$GRAFT_REPO_ROOT; TAG=${1:-cfg2}; shift
O=$R/gpurun_out/prof_$TAG
mkdir -p $O; rm -rf /tmp/prof_out
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 7 --warmup 2 "$@" > $O/bench_under_rocprof.log 2>&1
f=$(find /tmp/prof_out -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
t=$(find /tmp/prof_out -name "*kernel_trace.csv" | head -1)
python3 $R/tools/window_stats.py "$t" $O/bench_under_rocprof.log $O/kernel_stats_timed.csv
tail -1 $O/bench_under_rocprof.log | cut -c1-300
head -45 $O/kernel_stats_timed.csv
