#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/ubench/ubench.bin > gpurun_out/ubench.json 2> gpurun_out/ubench.err; echo "ubench rc=$?" )
cat gpurun_out/ubench.json
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python tools/gpu_sweep.py msmkind nttkind prover > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" )
cat gpurun_out/sweep.jsonl; tail -5 gpurun_out/sweep.err
