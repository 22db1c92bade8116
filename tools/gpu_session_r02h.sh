#!/bin/bash
mkdir -p gpurun_out
python tools/ntt_kinds.py > gpurun_out/ntt_adapt_on.json 2>gpurun_out/ntt_kinds.err; cat gpurun_out/ntt_adapt_on.json; echo
PLONK_NTT_ADAPTIVE_TILES=0 python tools/ntt_kinds.py > gpurun_out/ntt_adapt_off.json 2>>gpurun_out/ntt_kinds.err; cat gpurun_out/ntt_adapt_off.json; echo
python - <<'PY'
import sys,time,random,operator
sys.path.insert(0,'.')
from plonkathon_amd import _pypack
R=21888242871839275222246405745257275088548364400416034343698204186575808495617
keys=tuple("x%d"%i for i in range(2048))
def mk(i):
    vals,x=[],3+i
    for _ in range(2048):
        vals.append(x); x=x*x%R
    return dict(zip(keys,vals))
ws=[mk(i) for i in range(512)]
g=operator.itemgetter(*keys)
for rep in range(2):
    t=time.perf_counter(); b1=b"".join([_pypack.pack_le32(g(w),R) for w in ws]); d1=time.perf_counter()-t
    t=time.perf_counter(); b2=_pypack.pack_dicts_le32(ws,keys,R); d2=time.perf_counter()-t
    print(b1==b2, 'getter+pack %.1f us'%(d1/512*1e6), 'pack_dicts %.1f us'%(d2/512*1e6))
PY
