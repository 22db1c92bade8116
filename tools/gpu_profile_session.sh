#!/bin/bash
# kernel-time breakdowns: single-proof latency path (B=1) and the default bench shape
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b1 -o b1 -- python $R/bench.py --batch 1 --streams 1 --steps 20 --warmup 2 --no-cpu-baseline --no-microbench > $R/gpurun_out/prof_b1.log 2>&1; echo "b1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b256 -o b256 -- python $R/bench.py --batch 256 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $R/gpurun_out/prof_b256.log 2>&1; echo "b256 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "bench rc=$?"
cd $R
for d in prof_b1 prof_b256 prof_bench; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); echo "== $f"; cut -d, -f1-5 $f | sed -E 's/\(.*\)"/"/' | head -24; tail -1 gpurun_out/$d.log | cut -c1-200; done
