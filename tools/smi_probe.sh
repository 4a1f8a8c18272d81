#!/bin/bash
# sample power / clocks while the bench runs
( for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|fclk clock level|mclk clock level" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/smi_during.txt &
SMI=$!
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 1200 --warmup 5 > gpurun_out/bench_long.json 2>/dev/null
wait $SMI
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -4 > gpurun_out/smi_idle.txt
