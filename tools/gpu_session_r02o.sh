#!/bin/bash
# Chained column sums (PLONK_CHAIN), limb-wise conditional negation, multiplicative zero filter: parity suite, then
# A/B against the previous build (plonkathon_amd/libplonk_hip_prev.so = commit 8ee3b55), alternating, with the NTT
# micro-benchmarks on.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -3 gpurun_out/pytest_gpu.log
for rep in 1 2; do
  for lib in new prev; do
    if [ $lib = prev ]; then export PLONK_HIP_LIB=$R/plonkathon_amd/libplonk_hip_prev.so; else unset PLONK_HIP_LIB; fi
    ( timeout 600 python bench.py --steps 8 --no-cpu-baseline --no-fallbacks > gpurun_out/o_${lib}_${rep}.json 2> gpurun_out/o_${lib}_${rep}.err; echo "bench $lib $rep rc=$?" )
    grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/o_${lib}_${rep}.json | head -2 | tr '\n' ' '; echo
  done
done
for lib in new prev; do
  if [ $lib = prev ]; then export PLONK_HIP_LIB=$R/plonkathon_amd/libplonk_hip_prev.so; else unset PLONK_HIP_LIB; fi
  ( timeout 300 python bench.py --steps 6 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > gpurun_out/o_${lib}_1stream.json 2>/dev/null; echo "$lib 1 stream: $(grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/o_${lib}_1stream.json | head -2 | tr '\n' ' ')" )
done
unset PLONK_HIP_LIB
( timeout 120 ./tools/ubench/ubench.bin > gpurun_out/o_ubench.json 2>/dev/null; cut -c1-700 gpurun_out/o_ubench.json )
