#!/bin/bash
# VALU issue counters of a WHOLE step of the headline configuration (20 lock-step batches of 512 proofs on 20 streams, the 20-tooth
# comb): which kernel owns how many VALU instructions per proof, and what fraction of the step's issue slots they fill.
# One rocprofv3 --pmc pass (--kernel-trace only), then the same command unprofiled for the step's real duration (counter
# collection serialises dispatches: the profiled step is longer, its instruction counts are exact).
#     bash tools/pmc_step.sh <outdir> [extra bench.py args]      (step "stepvalu" of tools/gpu_session.sh)
out=${1:-gpurun_out/stepvalu}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/$out"
export TMPDIR=/tmp
cd /tmp
SIDE="--verify-samples 0 --no-cpu-baseline --no-microbench --no-fallbacks --no-end-to-end --no-configs --no-latency"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
    -d "$root/$out/step_issue" -o p -- python "$root/bench.py" --steps 1 --warmup 1 $SIDE --detail /tmp/step_detail_pmc.json "$@" \
    > "$root/$out/step_issue.json" 2> "$root/$out/step_issue.err"
echo "step_issue rc=$?"
timeout 600 python "$root/bench.py" --steps 10 --warmup 2 $SIDE --detail "$root/$out/step_plain_detail.json" "$@" \
    > "$root/$out/step_plain.json" 2> "$root/$out/step_plain.err"
echo "step_plain rc=$?"
cd "$root"
python tools/pmc_step_summary.py "$out" "$out/step_valu.json" | tail -40
