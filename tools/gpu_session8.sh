#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -k "msm or lincomb or k6 or commit or batch_prover_group" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python tools/gpu_sweep.py msm > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" )
grep -E '"M": 768|"M": 64' gpurun_out/sweep.jsonl; tail -5 gpurun_out/sweep.err
