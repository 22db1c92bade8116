#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or poly or batch_prover_k6 or 2_11 or poseidon or lagrange" > gpurun_out/pytest_ntt.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ntt.log )
python tools/ntt_kinds.py > gpurun_out/ntt_kinds_w4.json 2>gpurun_out/ntt_kinds.err; cat gpurun_out/ntt_kinds_w4.json
PLONK_HIP_LIB=$PWD/plonkathon_amd/libplonk_hip_w3.so python tools/ntt_kinds.py > gpurun_out/ntt_kinds_w3.json 2>>gpurun_out/ntt_kinds.err; cat gpurun_out/ntt_kinds_w3.json
for lib in libplonk_hip.so libplonk_hip_w3.so; do
PLONK_HIP_LIB=$PWD/plonkathon_amd/$lib timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benchc.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d.get('prover_ntt'))"
done
