# Dev tool (GPU box): config-2 step time cold / after a minute of load, with the shader clock logged by tools/probes/sclk_sampler.
tools/probes/sclk_sampler 150 > gpurun_out/sclk_thermal.log 2>&1 &
SP=$!
sleep 2
echo -n "cold: "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"
echo -n "load 60 s: "; python bench.py --no-cpu-baseline --steps 1300 --warmup 5 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"
echo -n "warm: "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"
echo -n "warm again: "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"
kill $SP
awk 'NR%10==1' gpurun_out/sclk_thermal.log | head -30
