cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w13
for v in 3 0 3 0; do echo -n "d_fp16_res=$v "; python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline --d-fp16-res $v 2>&1 | tail -1 | cut -c75-230; done | tee gpurun_out/w13/ab.log
