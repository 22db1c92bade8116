#!/bin/bash
# End-of-round evidence run on the GPU box: parity tests, bench line, rocprofv3 kernel stats, PMC passes, ubench.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-microbench > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof rc=$?" )
head -6 gpurun_out/prof_bench/bench_kernel_stats.csv | cut -c1-160
bash tools/pmc_collect.sh
( timeout 120 ./tools/ubench/gather.bin 160 > gpurun_out/gather.json 2>/dev/null; cat gpurun_out/gather.json )
( timeout 120 ./tools/ubench/ubench.bin > gpurun_out/ubench.json 2> gpurun_out/ubench.err; echo "ubench rc=$?"; head -c 600 gpurun_out/ubench.json )
