#!/bin/bash
# Default bench line of the final round-2 build (4 streams) and the rocprofv3 kernel trace of the same command (short).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cut -c1-300 gpurun_out/bench.json; echo; tail -2 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench4 -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench4.log 2>&1; echo "rocprof 4 streams rc=$?" )
grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/prof_bench4.log | head -3
( timeout 300 python -m pytest tests -m gpu -x -q -k "b512 or 512 or native" > gpurun_out/pytest_gpu_t.log 2>&1; tail -2 gpurun_out/pytest_gpu_t.log )
