cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w12
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "wino or full_width or n8 or spade or golden" > gpurun_out/w12/tests.log 2>&1; tail -3 gpurun_out/w12/tests.log
for v in 0 1 0 1; do echo -n "gmap=$v "; PG_WINO_GMAP=$v python bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' | tr '\n' ' '; echo; done | tee gpurun_out/w12/ab.log
