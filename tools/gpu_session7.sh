#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python tools/gpu_sweep.py nttkind prover > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" )
cat gpurun_out/sweep.jsonl; tail -5 gpurun_out/sweep.err
( timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof7 -o r01f -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-microbench > $GRAFT_REPO_ROOT/gpurun_out/prof7.log 2>&1; echo "rocprof rc=$?" )
head -12 gpurun_out/prof7/r01f_kernel_stats.csv | cut -c1-150
