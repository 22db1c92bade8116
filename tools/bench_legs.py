"""Side legs of bench.py: everything measured AFTER the timed region of the headline metric.

bench.py prints one short JSON line (the contract fields + `roofline` + `cpu_baseline`, under 4 KB so that the driver's
record keeps all of it); what the legs below measure goes into `bench_detail.json` beside it (and to stderr), a few scalars
of each into the line.  Every leg takes the `Run` namespace bench.py fills (contexts, provers, setup, program, sizes) and
returns a plain dictionary.  Nothing here is inside `value`.

  ClockSampler     shader clock / socket power over the timed region (rocm-smi)
  ubench_rates     ALU ceilings measured by tools/ubench on this round's build (profiles/rNN_ubench.json)
  valu_counters    SQ counter passes of this round's build (profiles/rNN_valu_summary.json, tools/pmc_valu.sh)
  fallbacks        the same prover on smaller MSM tables and on the bucket method; ec_lincomb on arbitrary bases
  end_to_end       a fresh pre-packed witness batch uploaded per lock-step batch inside a timed region
  poseidon         BASELINE configs[2] at group_order 2^10 / 2^11
  latency          ONE proof through both provers
  ntt_legs         BASELINE configs[3]: 2^16 .. 2^24 in both fields, the prover's sizes, MSMs/s; the 2^20 roofline block
  sampled_verify   a few random proofs of the last step under the pairing check (test.py:103-133 verifies what it proves)
  cpu_baseline     the oracle on one host core
"""
import glob
import json
import os
import re
import subprocess
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
NOMINAL_SCLK_MHZ = 2400.0
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BLS_PIN_NOTE = ("parity pinned BY DEFINITION only (O(n^2) DFT in Python integers + the published root of unity, tools/gen_bls_vectors.py): "
                "the reference has no BLS12-381 field (curve.py:2 imports py_ecc.bn128), so no reference-held vector can exist")


class Run:
    """What bench.py hands to the legs (plain attribute bag)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def latest_profile(suffix):
    """Newest committed profiles/rNN_<suffix> (by round number): (parsed JSON, repo-relative path) or (None, None)."""
    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)))
    if not paths:
        return None, None
    return json.load(open(paths[-1])), os.path.relpath(paths[-1], REPO)


def ubench_rates():
    """The bare-loop rates the ALU roofline is priced against, as tools/ubench measured them on THIS round's kernels headers
    (`ubench` step of tools/gpu_session.sh -> profiles/rNN_ubench.json).  No literal ceilings live in bench.py."""
    d, src = latest_profile("ubench.json")
    if not d:
        return None
    return {"g1_lazy_madd_G": d.get("g1_lazy_madd_Gops"), "fq_lazy_mul_G": d.get("fq_lazy_mul_Gops"), "fr_mul_G": d.get("fr_mul_Gops"),
            "fr_shoup_mul_G": d.get("fr_shoup_mul_Gops"), "clock_mhz": d.get("clock_mhz"), "source": src}


def valu_counters():
    d, src = latest_profile("valu_summary.json")
    if not d:
        return None
    d["source"] = src
    return d


class ClockSampler(threading.Thread):
    """Shader clock and socket power of this process's GPU while the timed region runs, from `rocm-smi --showclocks
    --showpower` every ~0.5 s (rocm-smi lists only the GPUs visible to the container; sysfs lists the whole node, and
    amdgpu's hwmon freq1_input is not the shader clock).  A separate short-lived process per sample: the prover's host
    thread is not touched.  Reports medians; None when rocm-smi is missing or prints nothing usable."""

    SCLK = re.compile(r"GPU\[(\d+)\].*sclk clock level:\s*\S+\s*\((\d+)Mhz\)")
    POWER = re.compile(r"GPU\[(\d+)\].*Power \(W\):\s*([\d.]+)")

    def __init__(self, index):
        super().__init__(daemon=True)
        self.index, self.samples, self.stop_flag = index, [], threading.Event()

    def run(self):
        while not self.stop_flag.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except (OSError, subprocess.SubprocessError):
                return
            f = [int(m.group(2)) for m in self.SCLK.finditer(out) if int(m.group(1)) == self.index]
            w = [float(m.group(2)) for m in self.POWER.finditer(out) if int(m.group(1)) == self.index]
            if not f:
                return
            self.samples.append((f[0], w[0] if w else None))
            self.stop_flag.wait(0.15)

    def summary(self):
        self.stop_flag.set()

        def med(xs):
            xs = sorted(x for x in xs if x is not None)
            return xs[len(xs) // 2] if xs else None

        fs = [a for a, _ in self.samples]
        if not fs:
            return None
        return {"sclk_mhz_median": med(fs), "sclk_mhz_min": min(fs), "sclk_mhz_max": max(fs),
                "socket_power_w_median": med([b for _, b in self.samples]), "samples": len(fs),
                "source": "rocm-smi --showclocks --showpower, one call every ~0.5 s over the timed region",
                "nominal_sclk_mhz": NOMINAL_SCLK_MHZ}


def lookup_table_bytes(n, c):
    windows = (255 + c - 1) // c
    return n * windows * (1 << (c - 1)) * 64 + n * (1 << (c - 1)) * 128  # table + one window of XYZZ staging


def comb_columns(h):
    return (254 + h - 1) // h


def comb_table_bytes(n, h):
    """csrc/msm.hip, msm_comb_bytes: the table and the XYZZ staging of its build (an eighth of the entries, at most 2^27)."""
    half, bases = 1 << (h - 1), max(n // 8, 1)
    while bases > 1 and bases * half > 1 << 27:
        bases //= 2
    return n * half * 64 + bases * half * 128


def comb_fit(n, budget):
    """The comb the library's automatic choice builds within `budget`: fewest columns, then fewest teeth."""
    fits = [h for h in range(8, 23) if comb_table_bytes(n, h) <= budget]
    best = min(comb_columns(h) for h in fits)
    return min(h for h in fits if comb_columns(h) == best)


# ----------------------------------------------------------------------------------------------------------------------
_NTT_SRC = {}


def ntt_microbench(ctx, log_n, batch, reps=5, inverse=False, in_place=False, field="bn254", profiled=False):
    """ms of one plonk_fr_ntt call (best of `reps`, HIP events on the library's stream) on `batch` transforms of 2^log_n.
    field = "bls12_381": plonk_bls_fr_ntt, the same kernels over the BLS12-381 scalar field (the buffer's 256-bit words are
    below both moduli: valid residues for either).  profiled: the library's per-pass events are recorded for exactly the
    `reps` timed calls (not the warm-up), and (best, mean) is returned."""
    import random

    from plonkathon_amd._lib import check

    ntt = ctx.L.plonk_bls_fr_ntt if field == "bls12_381" else ctx.L.plonk_fr_ntt
    n = 1 << log_n
    if id(ctx) not in _NTT_SRC:  # device-side fill: upload one random block and replicate it (content does not affect timing)
        rng = random.Random(12)
        _NTT_SRC[id(ctx)] = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])
    src = _NTT_SRC[id(ctx)]
    buf = ctx.alloc(n * batch)
    for off in range(0, n * batch, 4096):
        check(ctx.L.plonk_mem_d2d(ctx.handle, buf.at(off), src.ptr, 32 * min(4096, n * batch - off)))
    out = buf if in_place else ctx.alloc(n * batch)
    inv = 1 if inverse else 0
    for _ in range(2):
        check(ntt(ctx.handle, buf.ptr, out.ptr, log_n, inv, batch))  # warm: tables + scratch
    ctx.sync()
    if profiled:
        ctx.profile_reset()
        ctx.profile(True)
    times = []
    for _ in range(reps):
        ctx.timer_start()
        check(ntt(ctx.handle, buf.ptr, out.ptr, log_n, inv, batch))
        times.append(ctx.timer_stop_ms())
    if profiled:
        ctx.profile(False)
        return min(times), sum(times) / len(times)
    return min(times)


def ntt_queue_microbench(ctx, log_n, queue=16, reps=5, field="bn254"):
    """`queue` independent lone transforms of 2^log_n (one input, `queue` distinct outputs) enqueued back to back between ONE
    event pair -> ms per transform (best of `reps`).  The difference to `fwd` (one transform between an event pair, where the
    device idles while the host prepares the call) is the host cost per C-ABI call that is NOT hidden behind device work."""
    from plonkathon_amd._lib import check

    ntt = ctx.L.plonk_bls_fr_ntt if field == "bls12_381" else ctx.L.plonk_fr_ntt
    n = 1 << log_n
    src = _NTT_SRC[id(ctx)]
    buf = ctx.alloc(n)
    for off in range(0, n, 4096):
        check(ctx.L.plonk_mem_d2d(ctx.handle, buf.at(off), src.ptr, 32 * min(4096, n - off)))
    outs = [ctx.alloc(n) for _ in range(queue)]
    check(ntt(ctx.handle, buf.ptr, outs[0].ptr, log_n, 0, 1))
    ctx.sync()
    best = None
    for _ in range(reps):
        ctx.timer_start()
        for o in outs:
            check(ntt(ctx.handle, buf.ptr, o.ptr, log_n, 0, 1))
        ms = ctx.timer_stop_ms() / queue
        best = ms if best is None or ms < best else best
    return best


def ntt_sweep(ctx, comm, world, pmc):
    """BASELINE configs[3] / SURVEY.md 8(d): N = 2^16 .. 2^24 on random scalars — forward and inverse, out of place and in
    place, one transform alone and the constant-work batch [2^24 / N][N] (poly.py:113-148).  Per row: ms (slowest rank),
    whole-job GF-elems/s, fraction of the HBM roofline on the algorithmic 64 N bytes, PMC traffic where a pass exists."""
    from plonkathon_amd import distributed as D

    rows = {}
    for log_n in (16, 18, 20, 22, 24):
        n = 1 << log_n
        entry = {}
        for name, inverse, in_place, batch in (("fwd", False, False, 1), ("inv", True, False, 1), ("fwd_in_place", False, True, 1),
                                                ("inv_in_place", True, True, 1), ("fwd_batched", False, False, (1 << 24) >> log_n)):
            if name == "fwd_batched" and batch == 1:
                continue
            ms = D.max_over_ranks(ntt_microbench(ctx, log_n, batch, inverse=inverse, in_place=in_place), comm)
            gbs = 64.0 * n * batch / (ms * 1e-3) / 1e9
            entry[name] = {"ms": ms, "batch": batch, "gf_elems_per_s": world * n * batch / (ms * 1e-3), "hbm_frac": gbs / HBM_PEAK_GBS}
        # sixteen lone transforms behind one another between ONE event pair: the device never waits for the host
        ms = D.max_over_ranks(ntt_queue_microbench(ctx, log_n), comm)
        entry["fwd_queue16"] = {"ms_per_transform": ms, "queue": 16, "gf_elems_per_s": world * n / (ms * 1e-3), "hbm_frac": 64.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "host_gap_ms_vs_fwd": entry["fwd"]["ms"] - ms}
        # the field the configs[3] metric is quoted on upstream (BLS12-381 Fr): the same kernels, plonk_bls_fr_ntt
        for name, inverse, batch in (("bls12_381_fwd", False, 1), ("bls12_381_inv", True, 1), ("bls12_381_fwd_batched", False, (1 << 24) >> log_n)):
            if name.endswith("batched") and batch == 1:
                continue
            ms = D.max_over_ranks(ntt_microbench(ctx, log_n, batch, inverse=inverse, field="bls12_381"), comm)
            entry[name] = {"ms": ms, "batch": batch, "gf_elems_per_s": world * n * batch / (ms * 1e-3),
                           "hbm_frac": 64.0 * n * batch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "parity": "pinned by definition only"}
        tr = pmc.get("ntt_2^%d" % log_n)
        if tr:
            entry["pmc_traffic_bytes"] = tr
            entry["pmc_traffic_over_algorithmic"] = tr / (64.0 * n)
        rows["2^%d" % log_n] = entry
    return rows


def msm_microbench(ctx, bases, n, batch, reps=3):
    """`batch` commitments of n random coefficients in one plonk_g1_msm call -> ms (best of reps)."""
    import ctypes
    import random

    from plonkathon_amd._lib import check

    rng = random.Random(7)
    src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])
    sc = ctx.alloc(n * batch + 4096)
    for off in range(0, n * batch + 4096, 4096):
        check(ctx.L.plonk_mem_d2d(ctx.handle, sc.at(off), src.ptr, 32 * 4096))
    xy, fl = ctypes.create_string_buffer(64 * batch), ctypes.create_string_buffer(batch)
    call = lambda: check(ctx.L.plonk_g1_msm(ctx.handle, bases.handle, sc.ptr, n, batch, n + 1, xy, fl))  # stride n+1: distinct vectors
    call()
    best = None
    for _ in range(reps):
        ctx.sync()
        ctx.timer_start()
        call()
        ms = ctx.timer_stop_ms()
        best = ms if best is None or ms < best else best
    return best


def ntt_legs(run, pmc, pmc_src, valu):
    """Standalone NTT and MSM rates; with N GPUs every rank runs a replica and the whole-job rate is
    N x (work of one replica) / (time of the slowest rank).  Returns (detail, roofline block of the 2^20 transform)."""
    from plonkathon_amd import distributed as D

    ctx, comm, world = run.ctx, run.comm, run.world
    small = {}
    for log_n, batch in ((10, 512), (10, 4096), (11, 512), (11, 2048), (12, 512), (13, 512)):  # the prover's sizes: n and 4n of configs[1] / configs[2]
        ms = D.max_over_ranks(ntt_microbench(ctx, log_n, batch), comm)
        small["2^%d_x%d" % (log_n, batch)] = {"ms": ms, "gf_elems_per_s": world * batch * (1 << log_n) / (ms * 1e-3),
                                               "hbm_frac": 64.0 * batch * (1 << log_n) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    sweep = ntt_sweep(ctx, comm, world, pmc.get("ntt", {}))
    ms_msm = D.max_over_ranks(msm_microbench(ctx, run.setup.device_bases(ctx), run.group_order, 4608), comm)
    detail = {"ntt": {"prover_sizes": small, "configs3": sweep, "replicas": world, "pmc_source": pmc_src, "bls12_381_ntt_parity": BLS_PIN_NOTE},
              "msm": {"msms_per_s_2^11_x4608": world * 4608 / (ms_msm * 1e-3), "ms_4608": ms_msm, "replicas": world}}
    # The 2^20 transform.  ms_lone = best of 8 lone calls, one event pair around each call (the two launches run back to back).
    # Its two launches one by one: HIP events recorded around EACH pass on the library's stream in a second set of 8 calls
    # (rocprofv3 --kernel-trace durations of the same transforms: profiles/).  A call with the per-pass events inside is longer
    # than a plain one — the event records sit between the two launches (`ms_call_with_pass_events`) — so the sum of the passes
    # is to be read against ms_lone, and the profiled call's own duration is only reported so that nobody has to guess.
    ms20 = D.max_over_ranks(ntt_microbench(ctx, 20, 1, reps=8), comm)
    ms20_prof, ms20_prof_mean = ntt_microbench(ctx, 20, 1, reps=8, profiled=True)
    pc, pr_ = ctx.profile_read("ntt_pass_columns"), ctx.profile_read("ntt_pass_rows")
    ctx.profile_reset()
    per_pass = {"columns_us": round(1e3 * pc[0] / max(pc[1], 1), 2), "rows_us": round(1e3 * pr_[0] / max(pr_[1], 1), 2),
                "launches_each": pc[1], "ms_call_with_pass_events": ms20_prof, "ms_call_with_pass_events_mean": ms20_prof_mean,
                "note": "per-pass means over 8 calls that carry an event pair around each pass; ms_lone = best of 8 calls without them"}
    ach = 64.0 * (1 << 20) / (ms20 * 1e-3) / 1e9  # per GPU
    tr20 = pmc.get("ntt", {}).get("ntt_2^20")
    q20 = sweep["2^20"]["fwd_queue16"]["ms_per_transform"]
    roof = {"kernel": "ntt_wavel_column_kernel + ntt_wavel_kernel (N = 2^20 = 2^10 x 2^10, two launches)", "bound": "hbm",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "per_pass_us": per_pass, "ms_lone": ms20, "ms_in_a_queue_of_16": q20,
            "frac_in_a_queue_of_16": 64.0 * (1 << 20) / (q20 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": tr20, "traffic_over_algorithmic": tr20 / (64.0 * (1 << 20)) if tr20 else None, "traffic_source": pmc_src,
            "traffic_note": "includes 80 N bytes of inter-pass twiddles read from the table in usage order (one "
                            "multiplication per element instead of two, a deliberate bytes-for-instructions trade; "
                            "plonk_ntt_set_table_budget(0) gives 2.07 x the algorithmic 64 N instead of 3.3 x and a 6 % slower transform)"}
    ub = ubench_rates()
    alu = {}
    if valu and valu.get("ntt_2^20") and valu["ntt_2^20"].get("valu_insts_per_element"):
        # every VALU instruction of the two passes (rocprofv3 --pmc SQ_INSTS_VALU) at the measured issue cost, on 1024 SIMDs at the nominal clock
        v = valu["ntt_2^20"]
        cpi = [e.get("cycles_per_valu_inst") for e in v.get("passes", {}).values() if e.get("cycles_per_valu_inst")]
        cpi = sum(cpi) / len(cpi) if cpi else 4.0
        floor_us = (1 << 20) * v["valu_insts_per_element"] / 64.0 * cpi / 1024.0 / NOMINAL_SCLK_MHZ
        alu.update({"valu_insts_per_element": v["valu_insts_per_element"], "cycles_per_valu_inst": cpi, "floor_us_at_nominal_clock": floor_us,
                    "frac_of_alu_floor": floor_us / (ms20 * 1e3), "valu_busy_lone_profiled": v.get("valu_busy"), "source": valu.get("source")})
        roof["valu"] = {"valu_busy": v.get("valu_busy"), "valu_insts_per_element": v["valu_insts_per_element"]}
    if ub and ub.get("fr_shoup_mul_G"):
        # the twiddle multiplications alone: ~9.5 N (two passes + the inter-pass factor) at the bare-loop Shoup rate of tools/ubench
        alu.update({"shoup_mul_G_per_s": ub["fr_shoup_mul_G"], "mults_per_element": 9.5,
                    "multiplications_only_floor_us": 9.5 * (1 << 20) / (ub["fr_shoup_mul_G"] * 1e9) * 1e6, "ubench_source": ub["source"]})
    if alu:
        roof["alu"] = alu
    detail["roofline_ntt"] = roof
    detail["ntt"].update({"ms_2^11_x512": small["2^11_x512"]["ms"], "ms_2^16": sweep["2^16"]["fwd"]["ms"], "ms_2^20": ms20,
                          "ms_2^24": sweep["2^24"]["fwd"]["ms"], "gf_elems_per_s_2^11_x2048": small["2^11_x2048"]["gf_elems_per_s"],
                          "gf_elems_per_s_2^20": world * (1 << 20) / (ms20 * 1e-3)})
    return detail, roof


# ----------------------------------------------------------------------------------------------------------------------
def fallbacks(run):
    """The same prover when the HBM for the big table is not available (the library's default budget — 1/16 of the device's memory —,
    4 GiB, 40 GB, 80 GB), on the
    bucket method (what `north_star` names: Pippenger, no table), and ec_lincomb on ARBITRARY bases (curve.py:38-44: no SRS,
    no table — `plonk_srs_load_affine` + the bucket method)."""
    from plonkathon_amd import BatchProver, Context

    B, n = run.B, run.group_order
    fb = {}
    hbm_total = run.ctx.mem_info()[1]
    fit = lambda budget: comb_fit(n, budget)
    wfit = lambda budget: max(c for c in range(8, 18) if lookup_table_bytes(n, c) <= budget)
    wits = [run.witness_for(idx) for idx in run.mine[:B]]
    # (explicit sizes: with an automatic choice a context would simply attach to the big table the headline built; the headline's
    # 157.6 GB table stays resident meanwhile, so the legs here stay small — the 20-tooth comb of round 5 (68.7 GB + 17.2 GB of
    # staging, 13 additions per base) against the one with top tables is an A/B of its own: profiles/r06_m_comb_top_ab.json)
    for name, conf in (("library_default_budget", (0, fit(hbm_total // 16), hbm_total // 16)), ("table_budget_4GiB", (0, fit(4 << 30), 4 << 30)),
                       ("table_budget_1GiB", (0, fit(1 << 30), 1 << 30)), ("table_budget_128MiB", (0, fit(128 << 20), 128 << 20)),
                       ("window_table_40GB", (0, wfit(40e9), int(40e9), True)), ("window_table_default_budget", (0, wfit(hbm_total // 16), hbm_total // 16, True)),
                       ("bucket_method", (1, 0, 0))):
        c2 = Context(run.local_rank)
        c2.msm_lookup(*conf)
        pr = BatchProver(run.setup, run.program, c2)
        pr.upload(wits)
        for _ in range(2):
            pr.run()
            pr.download_raw()
        t = time.perf_counter()
        for _ in range(3):
            pr.run()
            st = pr.download_raw()[1]
        dt = (time.perf_counter() - t) / 3
        assert not any(st)
        i2 = run.setup.device_bases(c2).lookup_info()
        fb[name] = {"proofs_per_s": B / dt, "ms_per_batch_of_%d" % B: 1e3 * dt, "msm_table_layout": i2["layout"], "msm_table_bits": i2["bits"],
                    "additions_per_base": i2["additions_per_base"],
                    "msm_table_bytes": i2["bytes"], "msm_table_build_s": i2["build_s"],
                    "msm_table_fraction_of_hbm": i2["bytes"] / hbm_total, "fraction_of_value": (B / dt) / run.value, "streams": 1}
        del pr
        c2.close()
    if run.NS > 1:
        # the headline's own shape — one lock-step batch per stream, every stream busy — (a) on the LIBRARY-DEFAULT table budget: what a
        # caller who grants no memory gets from the same twenty streams; (b) on the bucket method: the algorithm `north_star` names
        # (Pippenger, no table at all) under the headline's conditions
        bits = fit(hbm_total // 16)
        for name, conf, groups in (("library_default_budget_all_streams", (0, bits, hbm_total // 16), 1), ("bucket_method_all_streams", (1, 0, 0), 0)):
            cs = [Context(run.local_rank) for _ in range(run.NS)]
            for c2 in cs:
                c2.msm_lookup(*conf)
                if groups:
                    c2.msm_configure(0, groups)  # (one workgroup per MSM, as the timed region: bench.py; the bucket method keeps its own choice)
            prs = [BatchProver(run.setup, run.program, c2) for c2 in cs]
            for k, q in enumerate(prs):
                q.upload(wits)  # (the same batch on every stream: the timing does not depend on the witnesses)

            def multi():
                for q in prs:
                    q.run()
                return [q.download_raw()[1] for q in prs]

            for _ in range(2):
                multi()
            t = time.perf_counter()
            for _ in range(3):
                sts = multi()
            dt = (time.perf_counter() - t) / 3
            assert not any(any(st) for st in sts)
            i2 = run.setup.device_bases(cs[0]).lookup_info()
            fb[name] = {"proofs_per_s": run.NS * B / dt, "ms_per_step": 1e3 * dt, "streams": run.NS, "msm_table_layout": i2["layout"],
                        "msm_table_bits": i2["bits"], "additions_per_base": i2["additions_per_base"], "msm_table_bytes": i2["bytes"],
                        "msm_table_fraction_of_hbm": i2["bytes"] / hbm_total, "fraction_of_value": (run.NS * B / dt) / run.value}
            del prs
            for c2 in cs:
                c2.close()
    fb["ec_lincomb_arbitrary_bases"] = ec_lincomb_arbitrary(run)
    return fb


def ec_lincomb_arbitrary(run, batch=1152):
    """`batch` MSMs over 2^11 ARBITRARY bases (random multiples of the generator loaded with plonk_srs_load_affine: not an SRS,
    so no lookup table — the bucket method, `north_star`'s Pippenger) in one plonk_g1_msm call, and one lone ec_lincomb."""
    import ctypes
    import random

    from plonkathon_amd import kzg
    from plonkathon_amd._lib import check

    n, ctx = run.group_order, run.ctx
    pts = run.setup.powers_of_x[:n]  # the same 2^11 points, handed over as plain affine coordinates: the library sees no SRS
    xy = b"".join(int(p[0]).to_bytes(32, "little") + int(p[1]).to_bytes(32, "little") for p in pts)
    h = ctypes.c_void_p()
    check(ctx.L.plonk_srs_load_affine(ctx.handle, xy, n, ctypes.byref(h)))
    bases = kzg._DeviceBases(ctx, h, n)
    ms = msm_microbench(ctx, bases, n, batch)
    out = {"msms_per_s_x%d" % batch: batch / (ms * 1e-3), "ms_x%d" % batch: ms, "msm_table_bits": bases.lookup_bits,
           "equivalent_proofs_per_s_msm_only": batch / 9.0 / (ms * 1e-3)}
    rng = random.Random(5)
    pairs = [(p, rng.randrange(R_MOD)) for p in pts]
    kzg.ec_lincomb(pairs)
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        kzg.ec_lincomb(pairs)
    out["lone_ec_lincomb_2^11_ms_incl_python_marshalling"] = 1e3 * (time.perf_counter() - t) / reps
    return out


def end_to_end(run):
    """What a caller who produces witnesses natively gets: every lock-step batch of a step is uploaded afresh inside the
    timed region (32 MiB per 512 proofs at 2^11, pre-packed in page-locked memory), the copy on the context's copy
    stream overlapping the other streams' rounds.  Same witnesses, same kernels, same downloads as `value`."""
    provers, parts, blobs = run.provers, run.parts, run.blobs
    V = len(provers[0].variables)
    pinned = []
    for pr, part, blob in zip(provers, parts, blobs):
        buf = pr.ctx.host_alloc(32 * V * len(part))
        buf[: len(blob)] = blob
        pinned.append(buf)

    def e2e_step():
        for pr, buf, part in zip(provers, pinned, parts):
            pr.upload_values_async(buf, len(part))   # H2D on the copy stream, conversion + gather behind an event
            pr.run()
        st = b"".join(pr.download_raw()[1] for pr in provers)
        assert not any(st)

    for _ in range(2):
        e2e_step()
    run.barrier()
    t0 = time.perf_counter()
    e2e_steps = max(3, min(run.steps, 5))
    for _ in range(e2e_steps):
        e2e_step()
    run.barrier()
    e2e = time.perf_counter() - t0
    for pr, buf in zip(provers, pinned):
        pr.ctx.host_free(buf)
    return {"proofs_per_s": e2e_steps * run.per_gpu / e2e, "ms_per_step": 1e3 * e2e / e2e_steps, "steps": e2e_steps,
            "fraction_of_value": (e2e_steps * run.per_gpu / e2e) / run.value, "uploaded_bytes_per_proof": 32 * V,
            "note": "a fresh pre-packed batch per lock-step batch inside the timed region: plonk_prover_upload_variables_async "
                    "from page-locked memory on a copy stream, overlapped with the other streams' rounds; witness generation "
                    "and packing (the caller's side) are outside, `host` has their Python cost"}


def poseidon(run, lines, proof_matches_fixture):
    """BASELINE configs[2]: the mini-Poseidon circuit (test.py:216-239; 1012 constraints) at the reference's own
    group_order 2^10 (test.py:250) and at 2^11; a lock-step batch of distinct witnesses (inputs (1, 2), (2, 3), ..)."""
    from plonkathon_amd import BatchProver, Program

    ctx, ctxs, NS = run.ctx, run.ctxs, run.NS
    cfg = {}
    PB = min(run.B, 512)
    for n_p in (1024, 2048):
        prog = Program(lines, n_p)
        t0 = time.perf_counter()
        wits = [prog.fill_variable_assignments({"L0": 1 + i, "M0": 2 + i}) for i in range(PB)]
        t_wit = time.perf_counter() - t0
        pr = BatchProver(run.setup, prog, ctx)
        pr.upload(wits)
        for _ in range(2):
            pr.run()
            pr.download_raw()
        reps = 5
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            pr.run()
            raw, st = pr.download_raw()
        dt = (time.perf_counter() - t0) / reps
        assert not any(st)
        cfg["poseidon_group_order_%d" % n_p] = {
            "proofs_per_s": PB / dt, "ms_per_batch_of_%d" % PB: 1e3 * dt, "constraints": len(lines), "streams": 1,
            "witness_generation_ms_per_proof": 1e3 * t_wit / PB,
            "proof_0_bit_identical_to_fixture": proof_matches_fixture(BatchProver.decode(raw[:768]), "poseidon_%d" % n_p)}
        del pr
        if n_p == 2048 and NS > 1:
            # the same circuit the way the headline runs: one lock-step batch of distinct witnesses per stream, all streams busy
            t0 = time.perf_counter()
            more = [prog.fill_variable_assignments({"L0": 1 + i, "M0": 2 + i}) for i in range(PB, NS * PB)]
            t_wit += time.perf_counter() - t0
            allw = wits + more
            prs = [BatchProver(run.setup, prog, c) for c in ctxs]
            for k, q in enumerate(prs):
                q.upload(allw[k * PB:(k + 1) * PB])

            def multi():
                for q in prs:
                    q.run()
                return [q.download_raw() for q in prs]

            for _ in range(2):
                multi()
            run.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                outs = multi()
            dtm = (time.perf_counter() - t0) / reps
            assert not any(any(st) for _, st in outs)
            cfg["poseidon_group_order_2048_all_streams"] = {"proofs_per_s": NS * PB / dtm, "ms_per_step": 1e3 * dtm, "streams": NS, "proofs_per_step": NS * PB,
                                                            "witness_generation_ms_per_proof": 1e3 * t_wit / (NS * PB)}
            del prs
    return {"configs[2]": cfg,
            "note": "one stream, one lock-step batch resident (the headline runs 20 batches on all its streams), and — "
                    "`_all_streams`, group_order 2^11 — one batch per stream of the headline's configuration; fixture = "
                    "tests/golden/oracle_proofs.json"}


def latency(run):
    """ONE proof of the configs[1] circuit (north_star: ">= 1000x reference-CPU proof-generation time"): through the
    reference's own entry point Prover(setup, program).prove(witness), with and without its sanity asserts
    (prover.py:108-116, 132-146, 205-219, 265-267, 288, 299), and through the lock-step prover with a batch of one."""
    from plonkathon_amd import BatchProver, Prover

    ctx = run.ctx

    def lat(fn, reps=10):
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ctx.sync()
            ts.append(time.perf_counter() - t)
        ts.sort()
        return {"best_ms": 1e3 * ts[0], "median_ms": 1e3 * ts[len(ts) // 2], "reps": reps}

    api = Prover(run.setup, run.program)
    wit0 = run.witness_for(run.mine[0])
    lt = {"api_prover_with_asserts": lat(lambda: api.prove(dict(wit0)))}
    api.check = False
    lt["api_prover"] = lat(lambda: api.prove(dict(wit0)))
    b1 = BatchProver(run.setup, run.program, ctx)
    lt["batch_prover_b1"] = lat(lambda: b1.prove(dict(wit0)))
    flat_a, flat_b = api.prove(dict(wit0)).flatten(), b1.prove(dict(wit0)).flatten()
    lt["api_equals_batch"] = all(flat_a[k] == flat_b[k] for k in flat_a)
    lt["proof_bytes"] = len(api.prove(dict(wit0)).to_bytes())
    lt["note"] = "wall time of one prove() call incl. witness staging and the download of the proof, warm (tables, Lagrange SRS and kernels loaded)"
    del b1
    return lt


def sampled_verify(run, gathered, total, k=4, seed=None):
    """`k` random proofs of the LAST step under the verifier's pairing check (VerificationKey.verify_proof: group arithmetic on
    the GPU, pairing product on the host, ~0.15 s each) — the reference verifies what it proves (test.py:103-133).  Untimed."""
    import random

    from plonkathon_amd import BatchProver

    rng = random.Random(seed if seed is not None else total)
    idx = sorted(rng.sample(range(total), min(k, total)))
    vk = run.setup.verification_key(run.program.common_preprocessed_input())
    ok = []
    for i in idx:
        proof = BatchProver.decode(gathered[i])
        wit = run.witness_for(i)
        public = [wit[v] for v in run.program.get_public_assignments()]
        ok.append(bool(vk.verify_proof(run.group_order, proof, public)))
    return {"indices": idx, "verified": ok, "all": all(ok)}


def cpu_baseline(ptau, program_lines, group_order):
    """The oracle on one host core (the reference is single-threaded pure Python): one full proof of the same
    workload, and the path's primitives one by one (BASELINE.md §3 / SURVEY.md §8(d))."""
    import random

    from oracle.circuit import Program as OProgram
    from oracle.fr_poly import fft_ints
    from oracle.g1 import ec_lincomb
    from oracle.plonk_prover import Prover as OProver
    from oracle.srs import Setup as OSetup

    osetup = OSetup.from_file(ptau)
    prog = OProgram(program_lines, group_order)
    wit = prog.fill_variable_assignments({"x0": 3})
    prover = OProver(osetup, prog)
    t0 = time.perf_counter()
    proof = prover.prove(dict(wit))
    dt = time.perf_counter() - t0

    def best_of(fn, reps=3):
        ts = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return min(ts), ts

    rng = random.Random(11)
    prim = {}
    for log_n in (11, 13, 16):
        vals = [rng.randrange(R_MOD) for _ in range(1 << log_n)]
        prim["fft_2^%d_ms" % log_n] = 1e3 * best_of(lambda: fft_ints(vals))[0]
        prim["ifft_2^%d_ms" % log_n] = 1e3 * best_of(lambda: fft_ints(vals, True))[0]
    scal = [rng.randrange(R_MOD) for _ in range(group_order)]
    pts = osetup.powers_of_x[:group_order]
    prim["ec_lincomb_2^11_s"] = best_of(lambda: ec_lincomb(list(zip(pts, scal))))[0]
    prim["samples"] = 3
    prim["note"] = "best of 3; oracle/fr_poly.py (poly.py:113-148 restated) and oracle/g1.py (curve.py:38-111 restated), 1 core"
    # the same primitives by the oracle's C half (oracle/c/bn254_oracle.c: iterative in-place NTT, Jacobian double-and-add,
    # 4 x 64-bit Montgomery limbs, one core, gcc -O2): what a plain compiled single-threaded CPU implementation does — a
    # fairer yardstick for the kernels than pure Python.  Only the C call is timed, not the marshalling of Python ints.
    try:
        import ctypes

        from oracle import c_oracle

        L = c_oracle.lib()
        cprim = {}
        for log_n in (11, 13, 16, 20):
            n = 1 << log_n
            raw = b"".join(rng.randrange(R_MOD).to_bytes(32, "little") for _ in range(min(n, 4096))) * (n // min(n, 4096))
            buf = (ctypes.c_uint64 * (4 * n)).from_buffer_copy(raw)
            cprim["ntt_2^%d_ms" % log_n] = 1e3 * best_of(lambda: L.oracle_fr_ntt(buf, ctypes.c_uint(log_n), ctypes.c_int(0)))[0]
        pb = (ctypes.c_uint64 * (8 * group_order)).from_buffer_copy(
            b"".join(int(p[0]).to_bytes(32, "little") + int(p[1]).to_bytes(32, "little") for p in pts))
        sb = (ctypes.c_uint64 * (4 * group_order)).from_buffer_copy(b"".join(int(x).to_bytes(32, "little") for x in scal))
        out, ident = (ctypes.c_uint64 * 8)(), ctypes.c_int(0)
        cprim["g1_lincomb_2^11_ms"] = 1e3 * best_of(lambda: L.oracle_g1_lincomb(pb, sb, ctypes.c_size_t(group_order), out, ctypes.byref(ident)))[0]
        cprim["note"] = "oracle/c (iterative in-place NTT, Jacobian double-and-add; same results as poly.py:113-148 / curve.py:38-111), 1 core, best of 3, C call only"
        prim["c"] = cprim
    except Exception as exc:  # the C oracle is optional test infrastructure: the Python figures above stand on their own
        prim["c"] = {"error": repr(exc)}
    return dt, proof, prim
