#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o bench -- python $R/bench.py --steps 3 --warmup 1 --batches-per-step 2 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $R/gpurun_out/prof_d.log 2>&1; echo "rocprof rc=$?" )
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_d/bench_kernel_stats.csv')))
for r in rows[:22]:
    print(r['Name'].split('(')[0][:40].ljust(40), r['Calls'].rjust(5), '%10.1f'%(float(r['TotalDurationNs'])/1e3), '%9.1f'%(float(r['AverageNs'])/1e3))
PY
for s in 1 2; do timeout 600 python bench.py --steps 4 --warmup 1 --batches-per-step 8 --streams $s --no-cpu-baseline --no-microbench --no-fallbacks 2>>gpurun_out/benchd.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams',$s, d['value'], d['ms_per_step'])"; done
