#!/usr/bin/env python3
"""Summarises tools/pmc_step.sh: VALU instructions per proof BY KERNEL over a whole step of the headline configuration, and the
fraction of the step's issue slots they fill.     usage: python tools/pmc_step_summary.py <dir of the passes> <out.json>

Method.  `step_issue/` is one rocprofv3 --pmc pass over `bench.py --steps 1 --warmup 1` (20 streams, 20-tooth comb): every dispatch
of every lock-step batch with SQ_INSTS_VALU (wave instructions), SQ_ACTIVE_INST_VALU (quad-cycles the VALUs were active),
GRBM_GUI_ACTIVE.  Counter collection serialises dispatches, so durations of that pass say nothing about the step — its instruction
and active-cycle counts are exact.  One lock-step batch (`run`) = 512 proofs; the number of runs in the pass = launches of
`quotient_kernel` (one per run); per-run figures = the pass's totals by kernel / runs.
`step_plain.json` is the same command unprofiled: ms_per_step and the sampled shader clock.  Then
    step_valu_busy = 4 * sum_k SQ_ACTIVE_INST_VALU(k per run) * batches_per_step / (1024 SIMDs * ms_per_step * sclk)
(rocprofiler's VALUBusy over the step instead of over one kernel), and per kernel its share of those active cycles."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
SIMDS = 1024.0
BATCH = 512.0

per = {}
for p in glob.glob(os.path.join(src, "step_issue", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
        d = per.setdefault(key, {"dur_ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by = collections.OrderedDict()
for (did, name), d in sorted(per.items()):
    e = by.setdefault(name, collections.defaultdict(float))
    e["launches"] += 1
    for k, v in d.items():
        e[k] += v
runs = by.get("quotient_kernel", {}).get("launches", 0.0)
if not runs:
    sys.exit("pmc_step_summary: no quotient_kernel dispatch in %s/step_issue" % src)

ONE_OFF = ("msm_comb_fill", "msm_comb_top_fill", "msm_comb_top_base", "msm_comb_scale", "msm_table_kernel", "g1_batch_to_affine", "msm_comb_delta", "fq_rescale", "msm_comb_verify",
           "li_coset", "witness_scatter", "public_gather", "fr_to_mont", "fr_powers", "ntt_program_block", "__amd_rocclr")
plain = {}
try:
    plain = json.loads(open(os.path.join(src, "step_plain.json")).read().strip().splitlines()[-1])
except Exception as exc:  # the summary still carries the per-kernel table
    sys.stderr.write("pmc_step_summary: no unprofiled line (%r)\n" % (exc,))
detail = {}
try:
    detail = json.load(open(os.path.join(src, "step_plain_detail.json")))
except Exception:
    pass
ms_per_step = plain.get("ms_per_step")
batches = plain.get("config", {}).get("batches_per_step")
sclk = (detail.get("clocks") or {}).get("sclk_mhz_median") or plain.get("roofline", {}).get("sclk_mhz")

rows, tot_insts, tot_active = [], 0.0, 0.0
for name, e in by.items():
    if any(name.startswith(x) for x in ONE_OFF):
        continue  # set-up work outside the step (table build, circuit preprocessing, uploads)
    insts, active = e.get("SQ_INSTS_VALU", 0.0) / runs, e.get("SQ_ACTIVE_INST_VALU", 0.0) / runs
    tot_insts += insts
    tot_active += active
    rows.append({"kernel": name, "launches_per_run": e["launches"] / runs, "valu_insts_per_proof": insts / BATCH,
                 "valu_active_quad_cycles_per_run": active, "cycles_per_valu_inst": 4.0 * active / insts if insts else None,
                 "serialised_us_per_run": e["dur_ns"] / runs / 1e3})
for r in rows:
    r["share_of_step_valu_active"] = r["valu_active_quad_cycles_per_run"] / tot_active if tot_active else None
rows.sort(key=lambda r: -r["valu_active_quad_cycles_per_run"])
res = {
    "method": __doc__.split("Method.")[1].strip(),
    "runs_in_the_pass": runs,
    "kernels": rows,
    "valu_wave_insts_per_proof": tot_insts / BATCH,
    "valu_active_quad_cycles_per_run": tot_active,
    "serialised_us_per_run": sum(r["serialised_us_per_run"] for r in rows),
    "ms_per_step": ms_per_step, "batches_per_step": batches, "sclk_mhz": sclk,
    "proofs_per_s_unprofiled": plain.get("value"),
}
if ms_per_step and batches and sclk:
    cycles = ms_per_step * 1e-3 * sclk * 1e6
    res["step_valu_busy"] = 4.0 * tot_active * batches / (SIMDS * cycles)
    for r in rows:
        r["ms_of_the_step_at_full_issue"] = 4.0 * r["valu_active_quad_cycles_per_run"] * batches / SIMDS / (sclk * 1e6) * 1e3
    res["ms_of_the_step_at_full_issue"] = sum(r["ms_of_the_step_at_full_issue"] for r in rows)
    res["note"] = ("step_valu_busy = the fraction of the step's %.1f ms in which a SIMD's VALU is executing, averaged over the 1024 SIMDs, at the "
                   "clock sampled in the unprofiled run; ms_of_the_step_at_full_issue per kernel = the time its VALU work takes with every "
                   "SIMD issuing back to back: what is left of the step is issue slots nobody filled" % ms_per_step)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k not in ("kernels", "method")}))
for r in rows[:14]:
    print("%-40s %6.2f launches  %10.0f insts/proof  %5.1f %%  %s ms" % (r["kernel"][:40], r["launches_per_run"], r["valu_insts_per_proof"],
          100 * (r["share_of_step_valu_active"] or 0), "%.2f" % r["ms_of_the_step_at_full_issue"] if "ms_of_the_step_at_full_issue" in r else "-"))
