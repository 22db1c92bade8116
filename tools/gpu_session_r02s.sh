#!/bin/bash
# How many HIP streams (contexts) per GPU: the 20 lock-step batches of a step dealt round-robin to 2 .. 10 streams.
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # streams
  timeout 400 python bench.py --steps 6 --streams $1 --no-cpu-baseline --no-microbench --no-fallbacks > gpurun_out/s_$1.json 2> gpurun_out/s_$1.err
  echo "streams=$1 rc=$? $(grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"sclk_mhz_median": [0-9.]*\|"socket_power_w_median": [0-9.]*' gpurun_out/s_$1.json | head -4 | tr '\n' ' ')"
}
run 2; run 4; run 5; run 6; run 8; run 10; run 4; run 2
