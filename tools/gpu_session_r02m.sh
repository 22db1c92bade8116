#!/bin/bash
# Lookup-kernel experiments: depth-one software pipeline of the table gather (default build) against the previous
# build, 3 and 2 waves per SIMD, and the timing probe whose gathers all hit cache (wrong results, upper bound).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -k "msm or lookup or commit or prove or batch" > gpurun_out/pytest_gpu_m.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_m.log )
tail -2 gpurun_out/pytest_gpu_m.log
run() {  # lib streams
  if [ $1 = main ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$R/plonkathon_amd/libplonk_hip_$1.so; fi
  timeout 400 python bench.py --steps 6 --streams $2 --no-cpu-baseline --no-microbench --no-fallbacks > gpurun_out/m_$1_$2.json 2> gpurun_out/m_$1_$2.err
  echo "$1 streams=$2 rc=$? $(grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/m_$1_$2.json | head -2 | tr '\n' ' ')"
}
run main 1; run prev 1; run w3 1; run w2 1; run nogather 1
run main 2; run prev 2; run w3 2; run nogather 2
run main 1; run prev 1
