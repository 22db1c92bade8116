#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q -k "k6 or ntt_vs_oracle or msm_vs_oracle" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 900 python tools/gpu_sweep.py msm ntt prover > gpurun_out/sweep.jsonl 2> gpurun_out/sweep.err; echo "sweep rc=$?" )
grep -E '"M": 768|"what": "ntt"|prover' gpurun_out/sweep.jsonl | grep -v '"batch": 1,' ; tail -5 gpurun_out/sweep.err
