#!/bin/bash
# Issue-slot view of the wave NTT kernels (rocprofv3 --pmc, SQ counters): what fraction of a wave's cycles issues VALU work,
# waits, or touches LDS, and whether the instruction cache keeps up (the kernels are straight-line code of 60-140 KB).
#   bash tools/pmc_ntt_util.sh <outdir>        (a step of tools/gpu_session.sh: "pmcu")
out=${1:-gpurun_out/pmcu}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail > "$root/$out/counters_available.txt" 2>&1 || rocprofv3 -L > "$root/$out/counters_available.txt" 2>&1
pass() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$root/$out/$name" -o p -- python "$root/tools/ntt_util.py" > "$root/$out/$name.log" 2>&1
  echo "$name rc=$?"
}
pass issue SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass icache SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH
pass mem SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + '/*/')):
    f = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not f:
        print(d, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r.get('Grid_Size', ''), r.get('Workgroup_Size', ''))
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    print('==', d)
    for k, v in agg.items():
        wc = v.get('SQ_WAVE_CYCLES', 0)
        if wc < 1e6 or 'ntt' not in k[0]:
            continue
        print(k, {c: round(x / wc, 4) for c, x in v.items() if c != 'SQ_WAVE_CYCLES'}, 'wave_cycles=%.3g' % wc)
PY
