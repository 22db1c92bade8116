#!/bin/bash
# Round-2 GPU session G: full parity suite, MSM reduction A/B (old LDS tree vs wave reduce), default bench line, rocprofv3 trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
for lib in libplonk_hip.so; do
PLONK_HIP_LIB=$PWD/plonkathon_amd/$lib timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benchg.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['host']['host_upload_ms_per_proof'])"
done
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json | cut -c1-1200; tail -3 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?" )
tail -2 gpurun_out/prof_bench.log | cut -c1-600
head -14 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-160
