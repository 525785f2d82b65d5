# Dev tool (GPU box): shader clock seen by a side process while bench.py runs (tools/probes/sclk_sampler.hip).
python bench.py --no-cpu-baseline --steps 300 --warmup 3 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
tools/probes/sclk_sampler ${1:-45} > gpurun_out/sclk.log 2>&1
wait $BP
cat gpurun_out/sclk.log
tail -1 gpurun_out/clk_bench.log | cut -c1-200
