#!/bin/bash
# Round-2 GPU session A: parity tests, ubench2 (FP64 / MFMA / cross-lane), default bench line, rocprofv3 kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -6 gpurun_out/pytest_gpu.log
( timeout 120 ./tools/ubench/ubench2.bin > gpurun_out/ubench2.json 2> gpurun_out/ubench2.err; echo "ubench2 rc=$?"; cat gpurun_out/ubench2.json )
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -25 gpurun_out/bench.err | grep -E "Elapsed|Maximum resident|Error|error|Traceback" 
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 2 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?" )
head -12 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-200
