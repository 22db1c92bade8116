#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/ubench/ubench.bin > gpurun_out/ubench.json 2> gpurun_out/ubench.err; echo "ubench rc=$?" )
cat gpurun_out/ubench.json
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -8 gpurun_out/pytest_gpu.log
( timeout 900 python tools/gpu_sweep.py msm prover ntt > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" )
grep -v '"M": 1,' gpurun_out/sweep.jsonl; tail -5 gpurun_out/sweep.err
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
