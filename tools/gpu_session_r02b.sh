#!/bin/bash
# Round-2 GPU session B: stream / batch / Lagrange sweeps of the prover bench (short runs).
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $*"; timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-microbench --no-fallbacks "$@" 2>>gpurun_out/sweepb.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'bits':d['config']['msm_table_bits'],'streams':d['config']['streams_per_gpu'],'B':d['config']['lockstep_batch'],'S':d['config']['batches_per_step'],'lag':d['config']['lagrange_commits'],'alu':d.get('roofline',{}).get('alu',{}).get('frac')}))
" | tee -a gpurun_out/sweepb.jsonl; }
rm -f gpurun_out/sweepb.jsonl
run --batches-per-step 8
run --batches-per-step 8 --streams 2
run --batches-per-step 8 --streams 4
run --batch 1024 --batches-per-step 4
run --batch 1024 --batches-per-step 4 --streams 2
run --batch 256 --batches-per-step 16 --streams 2
run --batches-per-step 8 --lagrange-commits --lookup-budget-gb 90
run --batches-per-step 8 --lookup-budget-gb 90
