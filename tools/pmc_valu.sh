#!/bin/bash
# Issue-slot and address-translation counters of the two kernels the roofline blocks are about:
#   msm_lookup_kernel (one lock-step batch of 512 proofs, one stream) and the two passes of a lone 2^20 transform,
# plus tools/ubench/gather.bin (random 64-byte reads over 4 MiB .. 160 GiB: read rate against table size = what address
# translation reach costs, with the UTCL1 hit / miss counts beside it).  One rocprofv3 --pmc pass per counter group,
# --kernel-trace only.      bash tools/pmc_valu.sh <outdir>       (step "valu" of tools/gpu_session.sh)
out=${1:-gpurun_out/valu}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/$out"
export TMPDIR=/tmp
cd /tmp
BENCH="$root/bench.py --steps 1 --warmup 1 --batch 512 --batches-per-step 1 --streams 1 --verify-samples 0 --no-cpu-baseline --no-microbench --no-fallbacks --no-end-to-end --no-configs --no-latency --detail /tmp/valu_detail.json"
pass() {  # name, command..., -- counters...
  name=$1; shift
  cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$root/$out/$name" -o p -- "${cmd[@]}" > "$root/$out/$name.log" 2>&1
  echo "$name rc=$?"
}
ISSUE="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
MEMI="SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
UTCL="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"
pass bench_issue python $BENCH -- $ISSUE
pass bench_mem python $BENCH -- $MEMI
pass bench_utcl python $BENCH -- $UTCL
pass ntt_issue python "$root/tools/ntt_only.py" 20 5 -- $ISSUE
pass ntt_mem python "$root/tools/ntt_only.py" 20 5 -- $MEMI
pass nttb_issue python "$root/tools/ntt_util.py" -- $ISSUE
pass gather_utcl "$root/tools/ubench/gather.bin" 160 -- $UTCL
timeout 300 "$root/tools/ubench/gather.bin" 160 > "$root/$out/gather_rates.json" 2> "$root/$out/gather_rates.err"
echo "gather rates rc=$?"
cd "$root"
python tools/pmc_valu_summary.py "$out" "$out/valu_summary.json" | tail -40
