#!/bin/bash
# First-contact GPU session: instruction-rate ubench, parity tests, bench, rocprof kernel stats.
# Usage (from the repo root on the GPU box):  bash tools/gpu_probe.sh
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 120 tools/ubench/ubench.bin > gpurun_out/ubench.json 2> gpurun_out/ubench.err; echo "ubench rc=$?" )
cat gpurun_out/ubench.json
( timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -15 gpurun_out/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log )
tail -3 gpurun_out/smoke.log
( timeout 600 python bench.py --steps 2 --warmup 1 --batch 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "rocprof rc=$?" )
ls -R gpurun_out/prof | head -20
