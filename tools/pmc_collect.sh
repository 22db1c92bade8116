#!/bin/bash
# PMC passes (one counter per pass, kernel-trace only) for the two dominant kernels.
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$ctr -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batch 512 --batches-per-step 1 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$ctr.log 2>&1
  echo "bench $ctr rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/ntt_$ctr -o p -- python $GRAFT_REPO_ROOT/tools/ntt_only.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/ntt_$ctr.log 2>&1
  echo "ntt $ctr rc=$?"
done
cd $GRAFT_REPO_ROOT
find gpurun_out/pmc -name "*.csv" | head -20
