#!/bin/bash
# Round-2 final evidence session: full parity suite, default bench line, rocprofv3 traces (2 streams and 1 stream), PMC passes.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -3 gpurun_out/pytest_gpu.log
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cut -c1-400 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof 2 streams rc=$?" )
grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/prof_bench.log | head -3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --batches-per-step 2 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench1.log 2>&1; echo "rocprof 1 stream rc=$?" )
grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/prof_bench1.log | head -3
bash tools/pmc_collect.sh > gpurun_out/pmc_collect.log 2>&1; tail -2 gpurun_out/pmc_collect.log
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r02_pmc_summary.json | grep -E "msm_lookup_kernel|ntt_2\^20|wave" | head
( timeout 120 ./tools/ubench/ubench2.bin > gpurun_out/ubench2.json 2>/dev/null; cut -c1-300 gpurun_out/ubench2.json )
