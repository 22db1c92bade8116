#!/bin/bash
# Round-2 evidence on the final build (token-threaded chain statements everywhere): full parity suite, default bench line
# with the rocm-smi clock sampler, 1 / 2 / 3 / 4 streams, rocprofv3 traces (2 streams and 1 stream), tools/ubench.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -3 gpurun_out/pytest_gpu.log
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cut -c1-300 gpurun_out/bench.json; echo; tail -2 gpurun_out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print('clocks', d.get('clocks'))
print('alu', {k:v for k,v in d['roofline']['alu'].items() if k!='note'})
PY
run() {  # streams
  timeout 400 python bench.py --steps 6 --streams $1 --no-cpu-baseline --no-microbench --no-fallbacks > gpurun_out/r_$1.json 2> gpurun_out/r_$1.err
  echo "streams=$1 rc=$? $(grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"sclk_mhz_median": [0-9.]*\|"socket_power_w_median": [0-9.]*' gpurun_out/r_$1.json | head -4 | tr '\n' ' ')"
}
run 1; run 2; run 3; run 4; run 2; run 3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 2 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof 2 streams rc=$?" )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --batches-per-step 2 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_bench1.log 2>&1; echo "rocprof 1 stream rc=$?" )
grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*' gpurun_out/prof_bench1.log | head -3
( timeout 120 ./tools/ubench/ubench.bin > gpurun_out/r_ubench.json 2>/dev/null; cut -c1-200 gpurun_out/r_ubench.json )
