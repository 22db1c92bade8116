#!/bin/bash
# Last session of round 2: the full parity suite on the final build (after the fix of the un-normalised negated y on the
# empty-accumulator path of g1l_madd_fast, found by the emulator build's range checks), then the default bench line.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 330 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -3 gpurun_out/pytest_gpu.log
( timeout 120 python bench.py --no-cpu-baseline --no-fallbacks > gpurun_out/bench_x.json 2> gpurun_out/bench_x.err; echo "bench rc=$?" )
cut -c1-260 gpurun_out/bench_x.json; echo
