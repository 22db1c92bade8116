#!/bin/bash
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or poly or distributed" > gpurun_out/pytest_ntt.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_ntt.log )
for lib in libplonk_hip_prev.so libplonk_hip.so; do
echo "== $lib"; PLONK_HIP_LIB=$PWD/plonkathon_amd/$lib python tools/ntt_kinds.py 2>>gpurun_out/ntt_kinds.err | tee gpurun_out/ntt_$lib.json
done
for lib in libplonk_hip_prev.so libplonk_hip.so libplonk_hip_prev.so libplonk_hip.so; do
PLONK_HIP_LIB=$PWD/plonkathon_amd/$lib timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benchj.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done
