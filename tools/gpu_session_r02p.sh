#!/bin/bash
# PLONK_CHAIN as token-threaded statements (default build) vs volatile statements (libplonk_hip_vol.so) vs the build
# without it (libplonk_hip_prev.so = commit 8ee3b55): quick parity subset, then alternating bench runs with the NTT
# micro-benchmarks.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or msm or prove" > gpurun_out/pytest_gpu_p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_p.log )
tail -2 gpurun_out/pytest_gpu_p.log
for rep in 1 2; do
  for lib in new vol prev; do
    if [ $lib = new ]; then unset PLONK_HIP_LIB; else export PLONK_HIP_LIB=$R/plonkathon_amd/libplonk_hip_$lib.so; fi
    ( timeout 600 python bench.py --steps 8 --no-cpu-baseline --no-fallbacks > gpurun_out/p_${lib}_${rep}.json 2> gpurun_out/p_${lib}_${rep}.err; echo "bench $lib $rep rc=$?" )
    python - <<PY
import json
d=json.load(open('gpurun_out/p_${lib}_${rep}.json'))
n=d['ntt']
print(round(d['value']), round(d['roofline']['avg_launch_us']), 'ntt 2^11x512 %.1f 2^11x2048 %.1f 2^13x512 %.1f G; 2^16 %.4f 2^20 %.4f ms; prover_ntt %.0f ms' % (n['gf_elems_per_s_2^11_x512']/1e9, n['gf_elems_per_s_2^11_x2048']/1e9, n['gf_elems_per_s_2^13_x512']/1e9, n['ms_2^16'], n['ms_2^20'], d['prover_ntt']['total_ms']))
PY
  done
done
