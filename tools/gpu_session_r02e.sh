#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt" > gpurun_out/pytest_ntt.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_ntt.log )
python tools/ntt_kinds.py > gpurun_out/ntt_kinds2.json 2>gpurun_out/ntt_kinds.err; cat gpurun_out/ntt_kinds2.json
for k in 4 0 4 0; do
timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --ntt-kind $k --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benche.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntt-kind',$k, d['value'], d['ms_per_step'])"; done
