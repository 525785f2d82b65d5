cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w15
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fused_up2 or upfirdn or modulated or full_width or n8 or golden" > gpurun_out/w15/tests.log 2>&1; tail -5 gpurun_out/w15/tests.log
for v in 0 1 0 1; do echo -n "up2_fused=$v "; PG_UP2_FUSED=$v python bench.py --no-cpu-baseline --conv-breakdown gpurun_out/w15/bd_$v.csv 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/w15/ab.log
grep "up2" gpurun_out/w15/bd_1.csv
