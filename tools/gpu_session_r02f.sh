#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 0 2 0 2 3 4; do
timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --msm-groups $g --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benchf.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('msm-groups',$g, d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
bash tools/pmc_collect.sh > gpurun_out/pmc_collect.log 2>&1; tail -3 gpurun_out/pmc_collect.log
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/r02_pmc_summary.json | head -30
