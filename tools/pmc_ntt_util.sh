#!/bin/bash
mkdir -p gpurun_out/pmcu
export TMPDIR=/tmp
cd /tmp
for kind in 0 1 2; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcu/ntt$kind -o p -- python $GRAFT_REPO_ROOT/tools/ntt_util.py $kind > $GRAFT_REPO_ROOT/gpurun_out/pmcu/ntt$kind.log 2>&1
echo "kind $kind rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcu/bench -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --batches-per-step 1 --streams 1 --no-cpu-baseline --no-microbench --no-fallbacks > $GRAFT_REPO_ROOT/gpurun_out/pmcu/bench.log 2>&1
echo "bench rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcu/*/')):
    f=glob.glob(d+'**/*counter_collection.csv',recursive=True)
    if not f: print(d,'no csv'); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in rows:
        k=(r['Kernel_Name'].split('(')[0], r.get('Grid_Size',''), r.get('Workgroup_Size',''))
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    print('==',d)
    for k,v in agg.items():
        if v.get('SQ_WAVE_CYCLES',0)<1e6: continue
        wc=v['SQ_WAVE_CYCLES']
        print(k, {c: round(x/wc,3) for c,x in v.items() if c!='SQ_WAVE_CYCLES'}, 'wave_cycles=%.3g'%wc)
PY
