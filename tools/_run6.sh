cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w6
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "streaming_head or one_channel_stencil or full_width or n8" > gpurun_out/w6/tests.log 2>&1; tail -3 gpurun_out/w6/tests.log
for rep in 1 2; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v; echo -n "head=$1 cin1=$2 "; PG_HEAD_STREAM=$1 PG_CIN1_STENCIL=$2 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c60-160; done; done | tee gpurun_out/w6/ab.log
