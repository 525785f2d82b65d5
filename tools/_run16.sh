cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w16
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "streaming_1x1 or two_source or full_width or n8 or spade or golden or fused_prologue" > gpurun_out/w16/tests.log 2>&1; tail -5 gpurun_out/w16/tests.log
for v in 0 1 0 1; do echo -n "s1x1=$v "; PG_S1X1=$v python bench.py --no-cpu-baseline --conv-breakdown gpurun_out/w16/bd_$v.csv 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/w16/ab.log
grep "^1,1,1" gpurun_out/w16/bd_1.csv | head -12
