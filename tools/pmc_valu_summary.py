#!/usr/bin/env python3
"""Summarises the SQ / TCP counter passes of tools/pmc_valu.sh into profiles/rNN_valu_summary.json — the file bench.py reads
`roofline.valu` and `roofline.secondary.valu` from.     usage: python tools/pmc_valu_summary.py <dir of the passes> <out.json>

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count QUAD-cycles, summed over all waves of the
launch; SQ_INSTS_* count wave-level instructions; GRBM_GUI_ACTIVE counts shader cycles, summed over the XCDs that report
it.  Derived per kernel (per launch, mean over the launches of the pass):
  cycles                 = GRBM_GUI_ACTIVE / xcd_factor  (xcd_factor in {1, 8}: the one that puts cycles / duration in 1.2 .. 2.6 GHz)
  effective_clock_ghz    = cycles / duration from the trace's own timestamps
  valu_busy              = 4 * SQ_ACTIVE_INST_VALU / (SIMDS * cycles)           [rocprofiler's VALUBusy; SIMDS = 1024]
  cycles_per_valu_inst   = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
  waves_per_simd         = 4 * SQ_WAVE_CYCLES / (SIMDS * cycles)                [average resident waves]
  wave_time_split        = VALU-active / waiting (s_waitcnt, barrier) / issue-stalled fractions of a wave's resident time
  valu_insts_per_addition (MSM: launch = 1152 MSMs x ADDS_PER_BASE x 2048 additions / 64 lanes: 24 876 per MSM on the comb of 21 teeth
  with top tables, 26 624 on the 20-tooth comb) and valu_insts_per_element (NTT)."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
SIMDS = 1024.0


def launches(tag):
    """{kernel: [ {counter: value, 'dur_ns': ..}, ... per dispatch ]}"""
    paths = glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True)
    per = collections.OrderedDict()
    for p in paths:
        for r in csv.DictReader(open(p)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size"]), int(r["Workgroup_Size"]))
            d = per.setdefault(key, {"dur_ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    by = collections.OrderedDict()
    for (did, name, grid, wg), d in sorted(per.items()):
        by.setdefault((name, grid, wg), []).append(d)
    return by


def mean(ds, skip_first=0):
    ds = ds[skip_first:] if len(ds) > skip_first else ds
    keys = set().union(*[set(d) for d in ds])
    return {k: sum(d.get(k, 0.0) for d in ds) / len(ds) for k in keys}, len(ds)


def derive(m):
    o = {k: m[k] for k in sorted(m)}
    gui, dur = m.get("GRBM_GUI_ACTIVE"), m.get("dur_ns")
    if gui and dur:
        fac = next((f for f in (1.0, 8.0, 32.0) if 1.2 <= gui / f / dur <= 2.6), None)
        if fac:
            cyc = gui / fac
            o.update({"xcd_factor": fac, "cycles": cyc, "effective_clock_ghz": cyc / dur})
            if "SQ_ACTIVE_INST_VALU" in m:
                o["valu_busy"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc)
            if "SQ_WAVE_CYCLES" in m:
                o["waves_per_simd"] = 4.0 * m["SQ_WAVE_CYCLES"] / (SIMDS * cyc)
    if m.get("SQ_INSTS_VALU") and "SQ_ACTIVE_INST_VALU" in m:
        o["cycles_per_valu_inst"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        o["wave_time_split"] = {k: m[c] / wc for k, c in (("valu_active", "SQ_ACTIVE_INST_VALU"), ("waiting", "SQ_WAIT_ANY"),
                                                          ("issue_stalled", "SQ_WAIT_INST_ANY"), ("lds_active", "SQ_ACTIVE_INST_LDS")) if c in m}
    if m.get("TCP_UTCL1_REQUEST_sum"):
        o["utcl1_miss_per_request"] = m.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / m["TCP_UTCL1_REQUEST_sum"]
        o["utcl1_hit_per_request"] = m.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0.0) / m["TCP_UTCL1_REQUEST_sum"]
    return o


res = {"units": __doc__.split("Units")[1].strip()[:1200]}

# ---- msm_lookup_kernel: every launch of the pass (2 prover runs x 4 launches: 1536 / 512 / 1536 / 1024 MSMs), counters SUMMED over
# the launches of a pass and set against the summed cycles / additions of the same launches (one run = 9 x 512 MSMs of 2^11
# scalars x 15 windows of 17 bits = 4608 x 30720 mixed additions; 64 lanes per wave instruction)
# (comb tables, csrc/msm_comb.h: msm_comb_kernel, 13 columns of 20 teeth = 4608 x 26624 mixed additions per run)
MSM_KERNEL = "msm_comb_kernel" if any(name == "msm_comb_kernel" for (name, _, _) in launches("bench_issue")) else "msm_lookup_kernel"
# (round 6, bench.py's default budget: 21 teeth + top tables = 12 columns of 2 073 real and virtual scalars = 24 876 additions per MSM
# of 2^11, 12.146 per base; MSM_ADDS_PER_BASE=13 for a pass on the 20-tooth comb, 15 for the 17-tooth one)
ADDS_PER_BASE = float(os.environ.get("MSM_ADDS_PER_BASE", str(24876 / 2048) if MSM_KERNEL == "msm_comb_kernel" else "15"))
ADDS_PER_RUN = 4608.0 * ADDS_PER_BASE * 2048
b = {}
for tag in ("bench_issue", "bench_mem", "bench_utcl"):
    ls = [d for (name, grid, wg), ds in launches(tag).items() if name == MSM_KERNEL for d in ds]
    if not ls:
        continue
    tot = collections.defaultdict(float)
    for d in ls:
        for k, v in d.items():
            tot[k] += v
    runs = len(ls) / 4.0
    adds = ADDS_PER_RUN * runs
    e = {"launches": len(ls), "prover_runs": runs, "sum": dict(tot)}
    fac = next((f for f in (1.0, 8.0, 32.0) if tot.get("GRBM_GUI_ACTIVE") and 1.2 <= tot["GRBM_GUI_ACTIVE"] / f / tot["dur_ns"] <= 2.6), None)
    if fac:
        cyc = tot["GRBM_GUI_ACTIVE"] / fac
        e.update({"xcd_factor": fac, "cycles": cyc, "effective_clock_ghz": cyc / tot["dur_ns"]})
        if "SQ_ACTIVE_INST_VALU" in tot:
            e["valu_busy"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc)
            e["valu_busy_per_launch"] = [round(4.0 * d["SQ_ACTIVE_INST_VALU"] / (SIMDS * d["GRBM_GUI_ACTIVE"] / fac), 4) for d in ls]
        if "SQ_WAVE_CYCLES" in tot:
            e["waves_per_simd"] = 4.0 * tot["SQ_WAVE_CYCLES"] / (SIMDS * cyc)
    if tot.get("SQ_INSTS_VALU"):
        e["valu_insts_per_addition"] = tot["SQ_INSTS_VALU"] / (adds / 64.0)
        e["cycles_per_valu_inst"] = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_INSTS_VALU"]
        # what the same instruction stream would do at one VALU instruction per SIMD every cycles_per_valu_inst cycles
        e["additions_per_s_at_full_issue_per_ghz"] = SIMDS * 1e9 / e["cycles_per_valu_inst"] * 64.0 / e["valu_insts_per_addition"]
    wc = tot.get("SQ_WAVE_CYCLES")
    if wc:
        e["wave_time_split"] = {k: tot[c] / wc for k, c in (("valu_active", "SQ_ACTIVE_INST_VALU"), ("waiting", "SQ_WAIT_ANY"),
                                                            ("issue_stalled", "SQ_WAIT_INST_ANY"), ("lds_active", "SQ_ACTIVE_INST_LDS")) if c in tot}
    for k, c in (("vmem_rd_insts_per_addition", "SQ_INSTS_VMEM_RD"), ("salu_insts_per_addition", "SQ_INSTS_SALU"), ("lds_insts_per_addition", "SQ_INSTS_LDS")):
        if c in tot:
            e[k] = tot[c] / (adds / 64.0)
    if tot.get("TCP_UTCL1_REQUEST_sum"):
        e["utcl1_requests_per_addition"] = tot["TCP_UTCL1_REQUEST_sum"] / adds
        e["utcl1_miss_per_request"] = tot.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0.0) / tot["TCP_UTCL1_REQUEST_sum"]
        e["utcl1_hit_per_request"] = tot.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0.0) / tot["TCP_UTCL1_REQUEST_sum"]
        if tot.get("GRBM_UTCL2_BUSY") and tot.get("GRBM_GUI_ACTIVE"):
            e["utcl2_busy_frac"] = tot["GRBM_UTCL2_BUSY"] / tot["GRBM_GUI_ACTIVE"]
    b[tag] = e
if b:
    top = {"passes": b, "note": "profiled launches run longer than unprofiled ones (counter collection serialises dispatches): compare ratios, not durations"}
    for k in ("valu_busy", "valu_busy_per_launch", "valu_insts_per_addition", "cycles_per_valu_inst", "waves_per_simd", "wave_time_split", "effective_clock_ghz",
              "additions_per_s_at_full_issue_per_ghz"):
        if k in b.get("bench_issue", {}):
            top[k] = b["bench_issue"][k]
    for k in ("utcl1_miss_per_request", "utcl1_requests_per_addition", "utcl2_busy_frac"):
        if k in b.get("bench_utcl", {}):
            top[k] = b["bench_utcl"][k]
    for k in ("vmem_rd_insts_per_addition", "salu_insts_per_addition"):
        if k in b.get("bench_mem", {}):
            top[k] = b["bench_mem"][k]
    top["additions_per_base"] = ADDS_PER_BASE
    res[MSM_KERNEL] = top

# ---- the two passes of a lone 2^20 transform
ntt = {}
for tag in ("ntt_issue", "ntt_mem"):
    for (name, grid, wg), ds in launches(tag).items():
        if "ntt_wavel" not in name:
            continue
        m, n = mean(ds, skip_first=1)
        e = ntt.setdefault(name, {"launches": n, "grid_threads": grid, "workgroup": wg})
        e.update({k: v for k, v in derive(m).items() if k not in e})
if ntt:
    tot_insts = sum(e.get("SQ_INSTS_VALU", 0.0) for e in ntt.values())
    tot_dur = sum(e.get("dur_ns", 0.0) for e in ntt.values())
    busy = sum(e.get("valu_busy", 0.0) * e.get("dur_ns", 0.0) for e in ntt.values()) / tot_dur if tot_dur else None
    res["ntt_2^20"] = {"passes": ntt, "valu_insts_per_element": tot_insts * 64.0 / (1 << 20), "valu_busy": busy,
                       "sum_of_pass_durations_us": tot_dur / 1e3}

# ---- the same kernels in steady state: batches of 2^8 .. 2^13 (the prover's shapes) and 16 x 2^20 (tools/ntt_util.py)
nb = {}
for (name, grid, wg), ds in launches("nttb_issue").items():
    if "ntt_wavel" not in name:
        continue
    m, n = mean(ds, skip_first=1)
    d = derive(m)
    elems = grid * {64: None, 256: None, 512: None, 1024: None}.get(wg, None) if False else None
    nb["%s grid=%d wg=%d" % (name, grid, wg)] = {k: d.get(k) for k in ("valu_busy", "waves_per_simd", "cycles_per_valu_inst", "effective_clock_ghz", "dur_ns", "wave_time_split")}
    nb["%s grid=%d wg=%d" % (name, grid, wg)]["launches"] = n
if nb:
    res["ntt_batched"] = nb

# ---- random 64-byte reads against the table size
g = launches("gather_utcl")
rates = {}
try:
    rates = json.load(open(os.path.join(src, "gather_rates.json")))
except Exception:
    pass
sizes = [k for k in rates if k.endswith("_GiB")]
gl = [d for (name, grid, wg), ds in g.items() if "k_gather" in name for d in ds]
if sizes:
    rows = {}
    for i, k in enumerate(sizes):
        ds = gl[3 * i:3 * i + 3]  # three launches per table size, in the order gather.bin runs them
        row = dict(rates[k])
        if ds:
            m, _ = mean(ds)
            d = derive(m)
            row.update({kk: d[kk] for kk in ("utcl1_miss_per_request", "utcl1_hit_per_request", "TCP_UTCL1_REQUEST_sum", "GRBM_UTCL2_BUSY", "cycles") if kk in d})
            if "GRBM_UTCL2_BUSY" in d and d.get("GRBM_GUI_ACTIVE"):
                row["utcl2_busy_frac"] = d["GRBM_UTCL2_BUSY"] / d["GRBM_GUI_ACTIVE"]
        rows[k] = row
    res["gather_random_64B"] = rows
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    if k != "units":
        print(k, json.dumps(v)[:1800])
