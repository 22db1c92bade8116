#!/bin/bash
# HBM traffic counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): one counter per pass, --kernel-trace only.
#   bash tools/pmc_collect.sh <outdir>       (a step of tools/gpu_session.sh: "pmc")
# Passes: the prover (one lock-step batch of 512, one stream), the lone NTTs of configs[3], and — FETCH_SIZE only — the
# random 64-byte gather microbenchmark whose byte count is known exactly: the calibration of FETCH_SIZE for the
# lookup MSM's access pattern (tools/pmc_summary.py).
out=${1:-gpurun_out/pmc}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/$out"
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$root/$out/bench_$ctr" -o p -- python "$root/bench.py" --steps 1 --warmup 1 --batch 512 --batches-per-step 1 --streams 1 --verify-samples 0 --detail /tmp/pmc_detail.json --no-cpu-baseline --no-microbench --no-fallbacks --no-end-to-end --no-configs --no-latency > "$root/$out/bench_$ctr.log" 2>&1
  echo "bench $ctr rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$root/$out/ntt_$ctr" -o p -- python "$root/tools/ntt_only.py" > "$root/$out/ntt_$ctr.log" 2>&1
  echo "ntt $ctr rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$root/$out/gather_FETCH_SIZE" -o p -- "$root/tools/ubench/gather.bin" 8 > "$root/$out/gather_FETCH_SIZE.log" 2>&1
echo "gather rc=$?"
cd "$root"
find "$out" -name "*counter_collection.csv" | head -20
python tools/pmc_summary.py "$out" "$out/pmc_summary.json" | tail -30
