#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof10 -o r01g -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof10.log 2>&1; echo "rocprof rc=$?" )
head -8 gpurun_out/prof10/r01g_kernel_stats.csv | cut -c1-160
bash tools/pmc_collect.sh
