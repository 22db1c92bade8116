#!/bin/bash
# Limb-form wave NTT kernel with its constants in scalar registers: full parity suite, kind 3 vs 5 micro-benchmark, the
# default bench line, bench with --ntt-kind 3 for the end-to-end A/B.
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log )
tail -3 gpurun_out/pytest_gpu.log
( NTT_KINDS=3,5 timeout 300 python tools/ntt_kinds.py > gpurun_out/v_ntt_kinds.json 2> gpurun_out/v_ntt_kinds.err; echo "ntt_kinds rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/v_ntt_kinds.json'))
for k in sorted(d):
    if k.startswith('kind3'):
        k5='kind5'+k[5:]
        print(k[6:], d[k], d.get(k5), 'ratio %.3f' % (d[k]['ms']/d[k5]['ms']))
PY
( timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" )
cut -c1-260 gpurun_out/bench.json; echo
run() {  # tag extra-args
  timeout 400 python bench.py --steps 6 --no-cpu-baseline --no-fallbacks $2 > gpurun_out/v_$1.json 2> gpurun_out/v_$1.err
  python - <<PY
import json
d=json.load(open('gpurun_out/v_$1.json'))
n=d['ntt']
print('$1', round(d['value']), 'ntt 2^11x512 %.1f 2^11x2048 %.1f 2^13x512 %.1f G; 2^16 %.4f 2^20 %.4f ms; prover_ntt %.0f ms' % (n['gf_elems_per_s_2^11_x512']/1e9, n['gf_elems_per_s_2^11_x2048']/1e9, n['gf_elems_per_s_2^13_x512']/1e9, n['ms_2^16'], n['ms_2^20'], d['prover_ntt']['total_ms']), 'host_upload', round(d['host']['host_upload_ms_per_proof'],4))
PY
}
run packed "--ntt-kind 3"; run limb ""; run packed2 "--ntt-kind 3"; run limb2 ""
