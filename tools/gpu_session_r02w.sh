#!/bin/bash
# Final round-2 evidence: NTT parity tests on the new dispatch (two-pass limb-form wave kernel for 2^18 / 2^20 / 2^22 at any
# batch), PMC passes of the standalone 2^20 transform, the default bench line, rocprofv3 traces at 4 streams and 1 stream.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests -m gpu -x -q -k "ntt or poly or two_pass or distributed" > gpurun_out/pytest_gpu_w.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_w.log )
tail -2 gpurun_out/pytest_gpu_w.log
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc/ntt_$ctr -o p -- python $R/tools/ntt_only.py > $R/gpurun_out/pmc/ntt_$ctr.log 2>&1
  echo "pmc ntt $ctr rc=$?"
done
cd $R
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cut -c1-260 gpurun_out/bench.json; echo
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench4 -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 8 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench4.log 2>&1; echo "rocprof 4 streams rc=$?" )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --batches-per-step 2 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench1.log 2>&1; echo "rocprof 1 stream rc=$?" )
grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/prof_bench1.log | head -3
find gpurun_out/pmc -name "*counter_collection.csv" | head
