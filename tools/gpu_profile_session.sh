#!/bin/bash
# kernel-time breakdowns: single-proof latency path (B=1) and the bench shape
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -o b1 -- python $R/bench.py --batch 1 --streams 1 --steps 20 --warmup 2 --no-cpu-baseline --no-microbench > $R/gpurun_out/prof_b1.log 2>&1; echo "b1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b512 -o b512 -- python $R/bench.py --batch 512 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $R/gpurun_out/prof_b512.log 2>&1; echo "b512 rc=$?"
cd $R
