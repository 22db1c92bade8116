#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -4 gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --batch 128 --dist-backend gloo --no-microbench > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench2 rc=$?" )
cat gpurun_out/bench2.json; tail -5 gpurun_out/bench2.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof11 -o r01h -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --batch 256 --no-cpu-baseline --no-microbench > $GRAFT_REPO_ROOT/gpurun_out/prof11.log 2>&1; echo "rocprof rc=$?" )
head -9 gpurun_out/prof11/r01h_kernel_stats.csv | cut -c1-60,150-260
