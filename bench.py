#!/usr/bin/env python3
"""bench.py — proofs/sec of the plonkathon prover hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or — when WORLD_SIZE is not set — bench.py spawns the
N ranks itself, one process per GPU.  Either way the ranks talk RCCL over xGMI through the library's own C-ABI
(`plonk_comm_*`, no PyTorch), and a run that cannot get N ranks on N GPUs fails instead of degrading.

A "step" is one pass of the hot path over one batch of synthetic input: `--batches-per-step` lock-step batches of
`--batch` independent PLONK proofs per GPU of the BASELINE configs[1] workload (group_order = 2^11, the powers-of-tau
SRS slice, synthetic witness — a 2047-gate squaring chain + one public input, ONE DISTINCT WITNESS PER PROOF).  With
the defaults a step is 20 x 512 = 10240 proofs per GPU (~0.3 s), so that the driver's 20 timed steps last > 5 s.
Proofs are independent, so N GPUs shard by proof index with no data-path collective; every step ends with the one
collective of the path, an all-gather of the finished proofs (768 B each) over RCCL ("scaling": "weak").  Inputs
(circuit polynomials, the MSM lookup table of the SRS, witness columns) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line: the contract fields, plus
  "roofline"      the dominant kernel of the timed region (msm_lookup; msm_accumulate if no table fits): durations from
                  HIP events recorded on the library's stream; `frac` is the kernel with the chip to itself (one stream
                  active, measured right after the timed region), `frac_concurrent` the same launches inside the
                  N-stream timed region; `traffic` from the committed PMC passes of this round's build;
  "roofline_ntt"  the standalone Fr NTT at 2^20 (BASELINE configs[3]) against the HBM roofline;
  "ntt"           BASELINE configs[3]: 2^16 / 18 / 20 / 22 / 24, forward and inverse, out of place and in place, a lone
                  transform and the constant-work batch [2^24 / N][N], each with GF-elems/s, the HBM-roofline fraction
                  and the PMC traffic, plus the same sizes over the BLS12-381 scalar field (`bls12_381_*`: plonk_bls_fr_ntt,
                  the field the upstream metric is quoted on); the prover's own sizes (2^10 .. 2^13, batched); N replicas
                  for N GPUs;
  "msm"           MSMs/s at 2^11;
  "configs"       BASELINE configs[2]: the mini-Poseidon circuit (test.py:216-239) at group_order 2^10 (the reference's
                  own size) and 2^11, a lock-step batch of distinct witnesses: proofs/s, and whether proof 0 — inputs
                  (1, 2) — is bit-identical to the committed fixture;
  "latency"       ONE proof of the configs[1] circuit: the reference-shaped `Prover(setup, program).prove(witness)` with
                  and without its sanity asserts, and `BatchProver.prove` (batch of one); speed-up over the oracle proof;
  "end_to_end"    the same step with a FRESH pre-packed batch uploaded for every lock-step batch inside the timed
                  region (page-locked host buffers, copy stream overlapped with the other streams' rounds);
  "fallbacks"     the same prover on the library's default 4 GiB table budget, on a 40 GB budget and on the bucket method;
  "host"          host-side cost of staging witnesses from Python dictionaries, end-to-end rate including it;
  "cpu_baseline"  the oracle (pure-Python port of the reference path) timed on this box, rank 0, N = 1: one full
                  proof, and per primitive fft/ifft at 2^11, 2^13, 2^16 and ec_lincomb at 2^11 (3 samples each).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails otherwise); the GPU boxes export this already —
# set before the HIP runtime is loaded in case a launcher scrubbed the environment
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
# chip-wide rate of the MSM loop's unit of work — the lazy mixed addition on register-resident operands —
# measured by tools/ubench in a burst of a few milliseconds, i.e. at the nominal 2.4 GHz shader clock
# (profiles/r02_q_ubench.json: g1_lazy_madd_Gops; fq_lazy_mul_Gops = 174).  Under the sustained prover load the
# chip holds ~2.07 GHz (`clocks` below), so `ceiling_at_sustained_clock` scales it by the measured clock.
G1_MADD_CEILING_G = 18.8
NOMINAL_SCLK_MHZ = 2400.0
MSM_WINDOW_BITS = 10      # bucket-method default (csrc/msm.hip); 26 windows of signed 10-bit digits
GROUP_ORDER = 2048
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")
DEFAULT_TABLE_GB = 150.0  # opt-in budget for the MSM lookup table: the c = 17 table of 2^11 bases is 128.8 GB + 17.2 GB of build staging


def chain_program_lines(n):
    """one public input + a squaring chain; every wire value is non-zero."""
    return ["x0 public"] + ["x%d <== x%d * x%d" % (i + 1, i, i) for i in range(n - 1)]


def witness_for(proof_index):
    """The witness dictionary fill_variable_assignments({"x0": 3 + index}) produces for the chain circuit."""
    vals, x = [], 3 + proof_index
    for _ in range(GROUP_ORDER):
        vals.append(x)
        x = x * x % R_MOD
    return dict(zip(["x%d" % i for i in range(GROUP_ORDER)], vals))


def poseidon_program_lines():
    """BASELINE configs[2]: the mini-Poseidon circuit generator of /root/reference/test.py:216-239 (round constants:
    test/poseidon_rc.json, kept as tests/golden/poseidon_rc.json; MDS = [1/3 .. 1/7], test/mini_poseidon.py:24):
    1012 constraints, public inputs L0, M0 and the hash M64."""
    rc = [[int(x) % R_MOD for x in row] for row in json.load(open(os.path.join(REPO, "tests", "golden", "poseidon_rc.json")))]
    mds = [pow(i, -1, R_MOD) for i in range(3, 8)]
    o = ["L0 public", "M0 public", "M64 public", "R0 <== 0"]
    for i in range(64):
        for j, pos in enumerate(("L", "M", "R")):
            if i < 4 or i >= 60 or pos == "L":
                o.append("%sadj%d <== %s%d + %d" % (pos, i, pos, i, rc[i][j]))
                o.append("%ssq%d <== %sadj%d * %sadj%d" % (pos, i, pos, i, pos, i))
                o.append("%sqd%d <== %ssq%d * %ssq%d" % (pos, i, pos, i, pos, i))
                o.append("%sqn%d <== %sqd%d * %sadj%d" % (pos, i, pos, i, pos, i))
            else:
                o.append("%sqn%d <== %s%d + %d" % (pos, i, pos, i, rc[i][j]))
        for j, pos in enumerate(("L", "M", "R")):
            o.append("%ssuma%d <== Lqn%d * %d" % (pos, i, i, mds[j]))
            o.append("%ssumb%d <== %ssuma%d + Mqn%d * %d" % (pos, i, pos, i, i, mds[j + 1]))
            o.append("%s%d <== %ssumb%d + Rqn%d * %d" % (pos, i + 1, pos, i, i, mds[j + 2]))
    return o


def proof_matches_fixture(proof, name):
    """Proof.flatten() against the committed fixture tests/golden/oracle_proofs.json (oracle proofs: the reference's
    rounds are blank upstream, the oracle is pinned by the reference's golden proof at group_order 8)."""
    case = [c for c in json.load(open(os.path.join(REPO, "tests", "golden", "oracle_proofs.json")))["cases"] if c["name"] == name][0]
    got = proof.flatten()
    for k, v in case["proof"].items():
        g = got[k]
        g = None if g is None else ([str(g[0].n), str(g[1].n)] if isinstance(g, tuple) else str(g.n))
        if g != v:
            return False
    return True


def lookup_table_bytes(n, c):
    windows = (255 + c - 1) // c
    return n * windows * (1 << (c - 1)) * 64 + n * (1 << (c - 1)) * 128  # table + one window of XYZZ staging


def cpu_baseline():
    """The oracle on one host core (the reference is single-threaded pure Python): one full proof of the same
    workload, and the path's primitives one by one (BASELINE.md §3 / SURVEY.md §8(d))."""
    import random

    from oracle.circuit import Program as OProgram
    from oracle.fr_poly import fft_ints
    from oracle.g1 import ec_lincomb
    from oracle.plonk_prover import Prover as OProver
    from oracle.srs import Setup as OSetup

    osetup = OSetup.from_file(PTAU)
    prog = OProgram(chain_program_lines(GROUP_ORDER), GROUP_ORDER)
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
    scal = [rng.randrange(R_MOD) for _ in range(GROUP_ORDER)]
    pts = osetup.powers_of_x[:GROUP_ORDER]
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
        pb = (ctypes.c_uint64 * (8 * GROUP_ORDER)).from_buffer_copy(
            b"".join(int(p[0]).to_bytes(32, "little") + int(p[1]).to_bytes(32, "little") for p in pts))
        sb = (ctypes.c_uint64 * (4 * GROUP_ORDER)).from_buffer_copy(b"".join(int(x).to_bytes(32, "little") for x in scal))
        out, ident = (ctypes.c_uint64 * 8)(), ctypes.c_int(0)
        cprim["g1_lincomb_2^11_ms"] = 1e3 * best_of(lambda: L.oracle_g1_lincomb(pb, sb, ctypes.c_size_t(GROUP_ORDER), out, ctypes.byref(ident)))[0]
        cprim["note"] = "oracle/c (iterative in-place NTT, Jacobian double-and-add; same results as poly.py:113-148 / curve.py:38-111), 1 core, best of 3, C call only"
        prim["c"] = cprim
    except Exception as exc:  # the C oracle is optional test infrastructure: the Python figures above stand on their own
        prim["c"] = {"error": repr(exc)}
    return dt, proof, prim


_NTT_SRC = {}


def ntt_microbench(ctx, log_n, batch, reps=5, inverse=False, in_place=False, field="bn254"):
    """ms of one plonk_fr_ntt call (best of `reps`, HIP events on the library's stream) on `batch` transforms of 2^log_n.
    field = "bls12_381": plonk_bls_fr_ntt, the same kernels over the BLS12-381 scalar field (the buffer's 256-bit words are
    below both moduli: valid residues for either)."""
    from plonkathon_amd._lib import check

    ntt = ctx.L.plonk_bls_fr_ntt if field == "bls12_381" else ctx.L.plonk_fr_ntt

    n = 1 << log_n
    import random

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
    best = None
    for _ in range(reps):
        ctx.timer_start()
        check(ntt(ctx.handle, buf.ptr, out.ptr, log_n, inv, batch))
        ms = ctx.timer_stop_ms()
        best = ms if best is None or ms < best else best
    return best


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


BLS_PIN_NOTE = ("parity pinned BY DEFINITION only (O(n^2) DFT in Python integers + the published root of unity, tools/gen_bls_vectors.py): "
                "the reference has no BLS12-381 field (curve.py:2 imports py_ecc.bn128), so no reference-held vector can exist")


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


def msm_microbench(ctx, bases, batch, reps=3):
    """`batch` commitments of 2^11 random coefficients in one plonk_g1_msm call -> ms (best of reps)."""
    import ctypes
    import random

    from plonkathon_amd._lib import check

    n = GROUP_ORDER
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


def spawn_ranks(args):
    """`--gpus N` without a launcher: one child process per GPU, rank 0's JSON line is ours."""
    import socket

    from plonkathon_amd import _lib
    import ctypes

    n_dev = ctypes.c_int(0)
    _lib.check(_lib.lib().plonk_device_count(ctypes.byref(n_dev)))
    if args.dist_backend == "rccl" and n_dev.value < args.gpus:
        sys.exit("bench.py: --gpus %d but only %d HIP device(s) are visible; refusing to run fewer ranks than asked "
                 "(use --dist-backend sockets to share a GPU between ranks for a functional test)" % (args.gpus, n_dev.value))
    with socket.socket() as s:  # a free port pair for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PLONK_RDZV_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        sys.exit("bench.py: rank exit codes %s" % rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="proofs per lock-step batch (BASELINE configs[4]: 512)")
    ap.add_argument("--batches-per-step", type=int, default=20, help="lock-step batches per GPU per step (all witnesses distinct)")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process (the HIP runtime maps its streams onto 4 hardware queues by default: more "
                         "compute streams then share them and lose the overlap they exist for); 0 = one per stream, at most 20, unless "
                         "the environment already sets it")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams per GPU (0 = one per lock-step batch of a step: --batches-per-step): the lock-step batches of a step are dealt round-robin to this many contexts, so one batch's latency-bound kernels (transcript, inversions, scans) overlap another's MSMs (measured 1 / 2 / 4 / 8 streams: 34.1 / 36.3 / 38.3 / 38.4 k proofs/s, profiles/r02_s_streams.txt)")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "sockets"],
                    help="transport of the final gather for N > 1: rccl = RCCL over xGMI through the C-ABI (default); sockets = TCP, lets ranks share one GPU")
    ap.add_argument("--lookup-budget-gb", type=float, default=DEFAULT_TABLE_GB,
                    help="HBM budget for the MSM lookup table (the library's own default is 4 GiB; the c = 17 table of 2^11 bases is 128.8 GB)")
    ap.add_argument("--force-comm", action="store_true",
                    help="with --gpus 1: still create a ONE-rank RCCL communicator and run the gather (plonk_gather_proofs_device), the "
                         "max over ranks and the barrier inside the timed region — the code path of an N-GPU run, exercised on one GPU")
    ap.add_argument("--no-lookup", action="store_true", help="bucket-method MSM only")
    ap.add_argument("--host-gather", action="store_true", help="N > 1 with RCCL: gather through host buffers (plonk_gather_results) instead of straight from the provers' device buffers (plonk_gather_proofs_device)")
    ap.add_argument("--dump-proofs", default="", help="rank 0 writes the last step's gathered proofs (768 bytes each, global order) to this file")
    ap.add_argument("--lagrange-commits", action="store_true", help="commit rounds 1-2 from Lagrange values over the Lagrange-basis SRS (a second lookup table)")
    ap.add_argument("--msm-groups", type=int, default=0, help="plonk_msm_configure groups: workgroups per MSM (0 = library default)")
    ap.add_argument("--ntt-kind", type=int, default=0, help="plonk_ntt_select_kernel: 0 auto, 1 / 4 the LDS kernel (radix-2 stages; A/B), 5 wave kernels wherever they apply, 6 / 7 wave kernels without / with the latency forms")
    ap.add_argument("--log-n", type=int, default=11, help="log2(group_order); 11 = the BASELINE workload, smaller values are for functional tests only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-fallbacks", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs[2] (Poseidon)")
    ap.add_argument("--no-latency", action="store_true")
    args = ap.parse_args()

    global GROUP_ORDER
    GROUP_ORDER = 1 << args.log_n
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # Before the HIP runtime initialises: one hardware queue per compute stream, at most 20.  The runtime's default of 4 makes
    # streams share queues (and serialise); more queues than the device schedules at once cost every latency-bound path.  Measured
    # (profiles/r04_j_streams_hw_queues.jsonl, same box within a session): 4 streams / 4 queues 40.1 - 40.3 k proofs/s, 8 / 16 41.1 k,
    # 12 / 24 41.4 k; another box: 8 / 16 42.0 k, 20 / 20 42.9 k, 20 / 40 43.1 k — but from 24 queues up the one-stream legs lose up to
    # half (Poseidon at 2^11 36.7 k -> 29.4 k at 24 queues, 17.7 k at 40; fresh uploads 0.98 -> 0.86 -> 0.78 of the headline)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues or min(20, max(4, args.streams or args.batches_per_step))))

    from plonkathon_amd import BatchProver, Context, Program, Setup, set_context
    from plonkathon_amd import distributed as D

    ctx = Context(local_rank)
    set_context(ctx)
    # librccl announces itself on C stdout ("RCCL version : ..", five lines, flushed whenever libc pleases — after this script's JSON
    # line when stdout is a pipe): while the communicator is created, file descriptor 1 points at stderr, and libc's buffer is flushed
    # before it is restored, so that stdout carries the one JSON line and nothing else
    import ctypes

    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        comm = D.init_from_env(ctx, args.dist_backend) if world > 1 else None
        if comm is None and args.force_comm:
            comm = D.RcclComm(ctx, 0, 1) if args.dist_backend == "rccl" else D.SocketComm(0, 1)
        if comm is not None and comm.kind == "rccl":
            comm.barrier()  # (the first collective: whatever the library prints lazily, it prints now)
            ctx.sync()
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    if comm is not None and comm.world != world:
        sys.exit("bench.py: communicator has %d ranks, expected %d" % (comm.world, world))
    budget = 0 if args.no_lookup else int(args.lookup_budget_gb * 1e9)
    B, S = args.batch, args.batches_per_step
    NS = max(1, args.streams or S)
    ctxs = [ctx] + [Context(local_rank) for _ in range(NS - 1)]
    for c in ctxs:
        c.msm_lookup(1 if args.no_lookup else 0, 0, budget)
        if args.msm_groups:
            c.msm_configure(0, args.msm_groups)
        if args.ntt_kind:
            from plonkathon_amd._lib import check as _check

            _check(c.L.plonk_ntt_select_kernel(c.handle, args.ntt_kind))
    setup = Setup.from_file(PTAU)
    program = Program(chain_program_lines(GROUP_ORDER), GROUP_ORDER)
    per_gpu = B * S
    total = per_gpu * world  # weak scaling: every GPU proves B * S proofs per step
    mine = D.shard_indices(total, rank, world)
    # S lock-step batches per step, dealt round-robin to NS contexts (HIP streams) of this GPU; every proof of a step has
    # its own witness (identical proofs would turn the MSM's table look-ups into cache hits), staged in HBM beforehand
    provers = [BatchProver(setup, program, ctxs[k % NS], lagrange_commits=args.lagrange_commits) for k in range(S)]
    parts = [mine[k * B : (k + 1) * B] for k in range(S)]
    # Witnesses: generated once per proof (Python), packed once to the device format ([B][V] x 32 B), uploaded; the packed
    # batches are kept for the `end_to_end` leg.  The dictionary route (BatchProver.upload: pack + copy + gather, synchronous)
    # is timed on the first batch.
    from plonkathon_amd.batch import _pack_witnesses
    t_gen = t_pack = t_up = 0.0
    blobs, host_upload_ms = [], None
    for k, (pr, part) in enumerate(zip(provers, parts)):
        t0 = time.perf_counter()
        wits = [witness_for(idx) for idx in part]
        t1 = time.perf_counter()
        if pr.variables:
            blob = _pack_witnesses(wits, pr.variables, R_MOD)
            t2 = time.perf_counter()
            pr.upload_values(blob, len(part))  # V x 32 B per proof -> HBM; wire columns gathered on the device
            t3 = time.perf_counter()
            blobs.append(blob)
        else:
            t2 = t1
            pr.upload(wits)
            t3 = time.perf_counter()
        t_gen += t1 - t0
        t_pack += t2 - t1
        t_up += t3 - t2
        if k == 0:
            t4 = time.perf_counter()
            pr.upload(wits)  # dicts -> bytes -> HBM in one call
            host_upload_ms = 1e3 * (time.perf_counter() - t4) / len(part)
    host_upload_packed_ms = 1e3 * t_up / per_gpu
    t_up = host_upload_ms * 1e-3 * per_gpu  # what staging every batch from dictionaries would cost (the `host` block's end-to-end figure)

    device_gather = comm is not None and comm.kind == "rccl" and not args.host_gather

    gather_ms = [0.0, 0.0, 0]  # this rank: collective ms, copy-to-host ms, gathers (reset before the timed region)

    def step():
        for pr in provers:
            pr.run()                   # five rounds + transcript: one stream of kernel launches each
        if device_gather:              # proofs go from the provers' device buffers into the all-gather, one host copy at the end
            gathered, status = D.gather_proofs_device(provers, B, total, comm)
            a_ms, h_ms = comm.last_gather_ms()   # HIP events around the ncclAllGather and the copy to the host
            gather_ms[0] += a_ms
            gather_ms[1] += h_ms
            gather_ms[2] += 1
            return gathered.parts[rank], status, gathered
        blobs = [pr.download_raw() for pr in provers]   # sync + 768 B per proof back to the host
        local = b"".join(b[0] for b in blobs)
        status = b"".join(b[1] for b in blobs)
        tg = time.perf_counter()
        gathered = D.gather_proofs_lazy(local, total, comm) if comm is not None else None  # the path's one collective
        if comm is not None:
            gather_ms[0] += 1e3 * (time.perf_counter() - tg)  # host wall time of the whole exchange (staging included)
            gather_ms[2] += 1
        return local, status, gathered

    def barrier():
        for c in ctxs:
            c.sync()
        if comm is not None:
            comm.barrier()

    for _ in range(args.warmup):
        proofs = step()
    for c in ctxs:
        c.profile_reset()
        c.profile(True)
    barrier()
    sampler = ClockSampler(local_rank) if rank == 0 else None  # one sampler per job: rank 0's GPU
    if sampler:
        sampler.start()
    gather_ms[:] = [0.0, 0.0, 0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proofs = step()
    for c in ctxs:
        c.sync()
    own_elapsed = time.perf_counter() - t0   # this rank's own clock, before it waits for the others
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, comm)
    clocks = sampler.summary() if sampler else None
    for c in ctxs:
        c.profile(False)

    def profile_sum(kernel):  # over every stream of this GPU
        parts = [c.profile_read(kernel) for c in ctxs]
        return sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts)

    assert not any(proofs[1]), "a proof in the batch reported a failure status"
    gathered = proofs[2] if comm is not None else D.gather_proofs_lazy(proofs[0], total, None)
    n_results = len(gathered)
    assert n_results == total and gathered.complete() and len(gathered[total - 1]) == 768
    if args.dump_proofs and rank == 0:
        with open(args.dump_proofs, "wb") as f:
            f.write(b"".join(gathered[i] for i in range(total)))

    # the dominant kernel: the lookup MSM when the table fits in HBM (default), else the bucket method's accumulate
    info = setup.device_bases(ctx).lookup_info()
    lookup_bits = info["bits"]
    msm_kernel = "msm_lookup" if lookup_bits else "msm_accumulate"
    msm_ms, msm_launches, msm_bytes = profile_sum(msm_kernel)
    # The same kernel with the chip to itself: with several streams a launch's event-to-event duration includes the time
    # it shares the CUs with the other streams' kernels, so the per-launch figures of the timed region understate the
    # kernel.  A short untimed phase runs the batches of stream 0 alone (the other streams idle) and reads its events.
    iso = None
    if NS > 1 and msm_launches:
        barrier()
        ctx.profile_reset()
        ctx.profile(True)
        for _ in range(max(3, -(-12 // len(provers[0::NS])))):  # (at least 48 launches of the kernel, however few batches stream 0 holds)
            for pr in provers[0::NS]:
                pr.run()
                pr.download_raw()
        ctx.sync()
        ctx.profile(False)
        i_ms, i_n, i_bytes = ctx.profile_read(msm_kernel)
        if i_n:
            iso = (i_ms * 1e-3 / i_n, i_bytes / i_n, i_n)
        barrier()
    total_proofs = args.steps * total
    line = {
        "metric": "proofs/sec at group_order=2^%d (PLONK prover hot path: NTT + quotient + KZG MSM)" % args.log_n,
        "value": total_proofs / elapsed,
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32x8 (254-bit Montgomery integers, BN254 Fr/Fq)",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: group_order=2^%d, powersOfTau28_hez_final_11 SRS slice, synthetic squaring-chain witness, one distinct witness per proof" % args.log_n,
            "proofs_per_gpu_per_step": per_gpu,
            "lockstep_batch": B,
            "batches_per_step": S,
            "prover": "BatchProver (lock-step, GPU-resident transcript)",
            "streams_per_gpu": NS,
            "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "results_gathered_per_step": n_results,
            "parallelism": "proof-sharded x%d" % world,
            "ranks_in_communicator": comm.world if comm is not None else 1,
            "gather_transport": comm.kind if comm is not None else "none (single rank)",
            "gather_in_timed_region": comm is not None,
            "gather_path": ("device buffers -> ncclAllGather -> host (plonk_gather_proofs_device)" if device_gather else
                            ("host buffers (plonk_gather_results / sockets)" if comm is not None else "none")),
            "msm_method": ("lookup table, %d-bit windows" % lookup_bits) if lookup_bits else "bucket method, %d-bit windows" % MSM_WINDOW_BITS,
            "msm_table_bits": lookup_bits,
            "msm_table_bytes": info["bytes"],
            "msm_table_build_s": info["build_s"],
            "msm_table_budget_bytes": budget,
            "msm_table_shared_by": info["sharers"],
            "msm_table_fraction_of_hbm": info["bytes"] / ctx.mem_info()[1],
            "lagrange_commits": bool(args.lagrange_commits),
            "timed_region_s": elapsed,
            "torch_imported": "torch" in sys.modules,  # the product and this file import no PyTorch: RCCL is reached through the C-ABI
        },
        "host": {
            "host_upload_ms_per_proof": host_upload_ms,
            "host_upload_prepacked_ms_per_proof": host_upload_packed_ms,
            "witness_generation_ms_per_proof": 1e3 * t_gen / per_gpu,
            "witness_packing_ms_per_proof": 1e3 * t_pack / per_gpu,
            "note": "host_upload: BatchProver.upload, Python witness dictionaries -> 32-byte words (V x 32 B per proof) -> HBM, wire "
                    "columns gathered on the device (timed on the first batch); prepacked: upload_values of already packed bytes "
                    "(all batches); both synchronous and outside `value` (inputs are resident before the timed region)",
        },
    }
    if comm is not None:
        # per-rank figures, so that a scaling record explains itself: every rank's own rate (its steps over its own clock, before
        # the barrier), the seconds its MSM table took to build, and the time of the step's one collective
        import struct

        mine_stats = struct.pack("<4d", args.steps * per_gpu / own_elapsed, info["build_s"],
                                 1e3 * gather_ms[0] / max(gather_ms[2], 1), 1e3 * gather_ms[1] / max(gather_ms[2], 1))
        rows = [struct.unpack("<4d", b[:32]) for b in comm.all_gather(mine_stats)]
        rates = [r[0] for r in rows]
        line["per_rank"] = {
            "proofs_per_s": rates, "proofs_per_s_min": min(rates), "proofs_per_s_max": max(rates), "proofs_per_s_sum": sum(rates),
            "msm_table_build_s": [r[1] for r in rows],
            "allgather_us_per_step": [r[2] for r in rows], "allgather_us_per_step_max": max(r[2] for r in rows),
            "gather_to_host_us_per_step": [r[3] for r in rows],
            "allgather_fraction_of_step": max(r[2] for r in rows) * 1e-6 / (elapsed / args.steps),
            "note": ("proofs_per_s: each rank's own steps over its own clock (value = all proofs over the slowest rank's clock, barrier "
                     "included); allgather_us_per_step: " + ("HIP events around the ncclAllGather of the step's proofs on the communicator's "
                     "stream, gather_to_host: the copy of all ranks' records to the host behind it" if device_gather else
                     "host wall time of the exchange through host buffers")),
        }
        if comm.kind == "rccl":
            ri = comm.info()
            line["config"]["rccl_path"], line["config"]["rccl_version"] = ri["path"], ri["version"]
            line["config"]["rccl_calls_issued"] = ri["collectives"]
    line["host"]["end_to_end_proofs_per_s_from_dicts_per_gpu"] = per_gpu / (t_up + per_gpu * elapsed / total_proofs * world)
    line["clocks"] = clocks  # rank 0's GPU; None when amdgpu's hwmon files are not visible

    # HBM bytes per launch / per transform from the committed PMC passes of this round's build (rocprofv3 cannot run
    # inside this process): profiles/r03_pmc_summary.json, written by tools/pmc_summary.py from tools/pmc_collect.sh
    pmc, pmc_src = {"bench": {}, "ntt": {}, "factors": {}}, None
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json"):
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            pmc, pmc_src = json.load(open(path)), "profiles/" + name
            break

    if msm_launches:
        avg_s = msm_ms * 1e-3 / msm_launches
        achieved = (msm_bytes / msm_launches) / avg_s / 1e9
        traffic = pmc["bench"].get(msm_kernel + "_kernel") if B == 512 else None
        line["roofline"] = {
            "kernel": msm_kernel + "_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --batch 512 --streams 1` on this "
                              "round's build (tools/pmc_collect.sh); FETCH_SIZE scaled by the factor calibrated for this access "
                              "pattern (`factors`), launch-weighted mean" % pmc_src,
            "traffic_factors": pmc.get("factors"),
            "launches": msm_launches,
            "avg_launch_us": avg_s * 1e6,
            "concurrent_streams": NS,
            "note": "algorithmic bytes = 96*N+64 per MSM (SURVEY.md 8(d)); the kernel is integer-ALU bound (DESIGN.md 3/4.2), "
                    "see `alu`; with the lookup table every addition also reads 64 table bytes, i.e. `traffic` is the real demand; "
                    "with concurrent_streams > 1 a launch shares the chip with the other streams' kernels, so avg_launch_us / achieved / "
                    "frac of the timed region are diluted by the concurrency: `isolated` is the same kernel with one stream active",
        }
        if iso:
            i_avg, i_bytes, i_n = iso
            msms_per_launch = i_bytes / (96.0 * GROUP_ORDER + 64.0)
            wb = lookup_bits or MSM_WINDOW_BITS
            i_gmadd = msms_per_launch * ((255 + wb - 1) // wb) * GROUP_ORDER / i_avg / 1e9
            line["roofline"]["isolated"] = {
                "avg_launch_us": i_avg * 1e6, "launches": i_n, "achieved": i_bytes / i_avg / 1e9,
                "frac": i_bytes / i_avg / 1e9 / HBM_PEAK_GBS, "g1_gmadd_per_s": i_gmadd,
                "alu_frac": i_gmadd / G1_MADD_CEILING_G,
                "alu_frac_at_sustained_clock": (i_gmadd / (G1_MADD_CEILING_G * clocks["sclk_mhz_median"] / NOMINAL_SCLK_MHZ)
                                                if clocks else None),
                "note": "the same kernel on the same inputs with one stream active (untimed phase right after the timed region): "
                        "its duration when it does not share the chip; profiles/ holds the rocprofv3 trace of a one-stream run"}
            # the headline figure is the kernel with the chip to itself; the N-stream figure of the timed region stays beside it
            r = line["roofline"]
            r["frac_concurrent"], r["achieved_concurrent"], r["avg_launch_us_concurrent"] = r["frac"], r["achieved"], r["avg_launch_us"]
            r["frac"], r["achieved"], r["avg_launch_us"] = r["isolated"]["frac"], r["isolated"]["achieved"], r["isolated"]["avg_launch_us"]
        if lookup_bits:  # the lookup method's own algorithmic bytes: one 64-byte table entry per addition + the scalars
            windows_l = (255 + lookup_bits - 1) // lookup_bits
            per_msm = GROUP_ORDER * windows_l * 64.0 + 32.0 * GROUP_ORDER + 64.0
            n_msm_l = msm_bytes / (96.0 * GROUP_ORDER + 64.0)
            line["roofline"]["method_bytes_per_msm"] = per_msm
            line["roofline"]["method_GBps"] = per_msm * n_msm_l / (msm_ms * 1e-3) / 1e9
            line["roofline"]["method_frac_of_peak"] = line["roofline"]["method_GBps"] / HBM_PEAK_GBS
        if traffic:  # what the kernel really asks of HBM (table look-ups), per the PMC passes
            line["roofline"]["traffic_GBps"] = traffic / (iso[0] if iso else avg_s) / 1e9
            line["roofline"]["traffic_frac_of_peak"] = line["roofline"]["traffic_GBps"] / HBM_PEAK_GBS
        # the honest ceiling: W*N mixed additions per MSM against the rate of a bare mixed-addition loop
        n_msm = msm_bytes / (96.0 * GROUP_ORDER + 64.0)
        wbits = lookup_bits or MSM_WINDOW_BITS
        windows = (255 + wbits - 1) // wbits
        gmadd = n_msm * windows * GROUP_ORDER / (msm_ms * 1e-3) / 1e9
        # whole-step view: every mixed addition of the timed region over its wall time — with several streams the per-launch
        # durations above include the time a launch shares the chip with another stream's kernels, this figure does not care
        step_gmadd = n_msm * windows * GROUP_ORDER / elapsed / 1e9
        sustained = G1_MADD_CEILING_G * clocks["sclk_mhz_median"] / NOMINAL_SCLK_MHZ if clocks else None
        line["roofline"]["alu"] = {"achieved_g1_gmadd_per_s": gmadd, "ceiling_g1_gmadd_per_s": G1_MADD_CEILING_G,
                                   "frac": gmadd / G1_MADD_CEILING_G,
                                   "whole_step_g1_gmadd_per_s": step_gmadd, "whole_step_frac": step_gmadd / G1_MADD_CEILING_G,
                                   "ceiling_at_sustained_clock": sustained,
                                   "whole_step_frac_at_sustained_clock": step_gmadd / sustained if sustained else None,
                                   "note": "%d mixed additions per MSM (8 Fq mul + 2 sqr + 8 add/sub each); ceiling = the same "
                                           "addition in a register-only loop (tools/ubench) timed in a millisecond burst at the "
                                           "nominal clock; the sustained prover load runs at the lower clock in `clocks`, and "
                                           "its additions also fetch 64 table bytes each" % (windows * GROUP_ORDER)}
    ntt_ms, ntt_launches, ntt_bytes = profile_sum("ntt_pass*")
    if ntt_launches:
        line["prover_ntt"] = {"kernel": "ntt passes inside the timed prover steps", "launches": ntt_launches, "total_ms": ntt_ms,
                              "achieved_GBps": ntt_bytes / (ntt_ms * 1e-3) / 1e9, "frac_of_hbm_peak": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if not args.no_fallbacks and lookup_bits and world == 1:
        # the same prover when the HBM for the big table is not available: a 40 GB budget, and no table at all
        fb = {}
        c75 = max(c for c in range(8, 18) if lookup_table_bytes(GROUP_ORDER, c) <= 80e9)
        c40 = max(c for c in range(8, 18) if lookup_table_bytes(GROUP_ORDER, c) <= 40e9)
        c4 = max(c for c in range(8, 18) if lookup_table_bytes(GROUP_ORDER, c) <= 4 << 30)
        hbm_total = ctx.mem_info()[1]
        for name, conf in (("library_default_4GiB", (0, c4, 4 << 30)), ("table_budget_40GB", (0, c40, int(40e9))),
                           ("table_budget_80GB", (0, c75, int(80e9))), ("bucket_method", (1, 0, 0))):
            c2 = Context(local_rank)
            c2.msm_lookup(*conf)
            pr = BatchProver(setup, program, c2)
            pr.upload([witness_for(idx) for idx in mine[:B]])
            for _ in range(2):
                pr.run()
                pr.download_raw()
            t = time.perf_counter()
            for _ in range(3):
                pr.run()
                st = pr.download_raw()[1]
            dt = (time.perf_counter() - t) / 3
            assert not any(st)
            i2 = setup.device_bases(c2).lookup_info()
            fb[name] = {"proofs_per_s": B / dt, "ms_per_batch_of_%d" % B: 1e3 * dt, "msm_table_bits": i2["bits"],
                        "msm_table_bytes": i2["bytes"], "msm_table_build_s": i2["build_s"],
                        "msm_table_fraction_of_hbm": i2["bytes"] / hbm_total, "fraction_of_value": (B / dt) / (total_proofs / elapsed), "streams": 1}
            del pr
            c2.close()
        line["fallbacks"] = fb
        for k, v in fb.items():  # scalars inside `config`, where the driver's record keeps them
            line["config"]["fallback_%s_proofs_s" % k] = round(v["proofs_per_s"], 1)

    if not args.no_end_to_end and world == 1 and provers[0].variables:
        # What a caller who produces witnesses natively gets: every lock-step batch of a step is uploaded afresh inside the
        # timed region (32 MiB per 512 proofs at 2^11, pre-packed in page-locked memory), the copy on the context's copy
        # stream overlapping the other streams' rounds.  Same witnesses, same kernels, same downloads as `value`.
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
        barrier()
        t0 = time.perf_counter()
        e2e_steps = max(3, min(args.steps, 5))
        for _ in range(e2e_steps):
            e2e_step()
        barrier()
        e2e = time.perf_counter() - t0
        line["end_to_end"] = {
            "proofs_per_s": e2e_steps * per_gpu / e2e, "ms_per_step": 1e3 * e2e / e2e_steps, "steps": e2e_steps,
            "fraction_of_value": (e2e_steps * per_gpu / e2e) / (total_proofs / elapsed),
            "uploaded_bytes_per_proof": 32 * V,
            "note": "a fresh pre-packed batch per lock-step batch inside the timed region: plonk_prover_upload_variables_async "
                    "from page-locked memory on a copy stream, overlapped with the other streams' rounds; witness generation "
                    "and packing (the caller's side) are outside, `host` has their Python cost"}
        line["config"]["end_to_end_fraction"] = line["end_to_end"]["fraction_of_value"]
        for pr, buf in zip(provers, pinned):
            pr.ctx.host_free(buf)

    if not args.no_configs and world == 1 and args.log_n == 11:  # (smaller --log-n values are functional tests of the launch contract)
        # BASELINE configs[2]: the mini-Poseidon circuit (test.py:216-239; 1012 constraints) at the reference's own
        # group_order 2^10 (test.py:250) and at 2^11; a lock-step batch of distinct witnesses (inputs (1, 2), (2, 3), ..)
        lines = poseidon_program_lines()
        cfg = {}
        PB = min(B, 512)
        for n_p in (1024, 2048):
            prog = Program(lines, n_p)
            t0 = time.perf_counter()
            wits = [prog.fill_variable_assignments({"L0": 1 + i, "M0": 2 + i}) for i in range(PB)]
            t_wit = time.perf_counter() - t0
            pr = BatchProver(setup, prog, ctx)
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
            line["config"]["poseidon_2^%d_proofs_s" % (n_p.bit_length() - 1)] = round(PB / dt, 1)
            line["config"]["poseidon_2^%d_proof_0_matches_fixture" % (n_p.bit_length() - 1)] = cfg["poseidon_group_order_%d" % n_p]["proof_0_bit_identical_to_fixture"]
            del pr
            if n_p == 2048 and NS > 1:
                # the same circuit the way the headline runs: one lock-step batch of distinct witnesses per stream, all streams busy
                t0 = time.perf_counter()
                more = [prog.fill_variable_assignments({"L0": 1 + i, "M0": 2 + i}) for i in range(PB, NS * PB)]
                t_wit += time.perf_counter() - t0
                allw = wits + more
                prs = [BatchProver(setup, prog, c) for c in ctxs]
                for k, q in enumerate(prs):
                    q.upload(allw[k * PB:(k + 1) * PB])

                def multi():
                    for q in prs:
                        q.run()
                    return [q.download_raw() for q in prs]

                for _ in range(2):
                    multi()
                barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    outs = multi()
                dtm = (time.perf_counter() - t0) / reps
                assert not any(any(st) for _, st in outs)
                cfg["poseidon_group_order_2048_all_streams"] = {"proofs_per_s": NS * PB / dtm, "ms_per_step": 1e3 * dtm, "streams": NS, "proofs_per_step": NS * PB,
                                                                "witness_generation_ms_per_proof": 1e3 * t_wit / (NS * PB)}
                line["config"]["poseidon_2^11_proofs_s_%d_streams" % NS] = round(NS * PB / dtm, 1)
                del prs
        line["configs"] = {"configs[2]": cfg,
                           "note": "one stream, one lock-step batch resident (the headline runs 20 batches on all its streams), and — "
                                   "`_all_streams`, group_order 2^11 — one batch per stream of the headline's configuration; fixture = "
                                   "tests/golden/oracle_proofs.json"}

    if not args.no_latency and world == 1:
        # ONE proof of the configs[1] circuit (north_star: ">= 1000x reference-CPU proof-generation time"): through the
        # reference's own entry point Prover(setup, program).prove(witness), with and without its sanity asserts
        # (prover.py:108-116, 132-146, 205-219, 265-267, 288, 299), and through the lock-step prover with a batch of one
        from plonkathon_amd import Prover

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

        api = Prover(setup, program)
        wit0 = witness_for(mine[0])
        lt = {"api_prover_with_asserts": lat(lambda: api.prove(dict(wit0)))}
        api.check = False
        lt["api_prover"] = lat(lambda: api.prove(dict(wit0)))
        b1 = BatchProver(setup, program, ctx)
        lt["batch_prover_b1"] = lat(lambda: b1.prove(dict(wit0)))
        flat_a, flat_b = api.prove(dict(wit0)).flatten(), b1.prove(dict(wit0)).flatten()
        lt["api_equals_batch"] = all(flat_a[k] == flat_b[k] for k in flat_a)
        lt["proof_bytes"] = len(api.prove(dict(wit0)).to_bytes())
        lt["note"] = "wall time of one prove() call incl. witness staging and the download of the proof, warm (tables, Lagrange SRS and kernels loaded)"
        line["latency"] = lt
        line["config"]["latency_api_ms"] = lt["api_prover"]["median_ms"]
        line["config"]["latency_api_with_asserts_ms"] = lt["api_prover_with_asserts"]["median_ms"]
        line["config"]["latency_batch_of_one_ms"] = lt["batch_prover_b1"]["median_ms"]
        del b1

    if not args.no_microbench:
        # standalone NTT and MSM rates; with N GPUs every rank runs a replica and the whole-job
        # rate is N x (work of one replica) / (time of the slowest rank)
        small = {}
        for log_n, batch in ((10, 512), (10, 4096), (11, 512), (11, 2048), (12, 512), (13, 512)):  # the prover's sizes: n and 4n of configs[1] / configs[2]
            ms = D.max_over_ranks(ntt_microbench(ctx, log_n, batch), comm)
            small["2^%d_x%d" % (log_n, batch)] = {"ms": ms, "gf_elems_per_s": world * batch * (1 << log_n) / (ms * 1e-3),
                                                   "hbm_frac": 64.0 * batch * (1 << log_n) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        sweep = ntt_sweep(ctx, comm, world, pmc.get("ntt", {}))
        ms_msm = D.max_over_ranks(msm_microbench(ctx, setup.device_bases(ctx), 4608), comm)
        line["ntt"] = {"prover_sizes": small, "configs3": sweep, "replicas": world,
                       "pmc_source": pmc_src,
                       # the round-2 keys, kept for comparisons across rounds
                       "ms_2^11_x512": small["2^11_x512"]["ms"], "ms_2^16": sweep["2^16"]["fwd"]["ms"], "ms_2^20": sweep["2^20"]["fwd"]["ms"],
                       "gf_elems_per_s_2^11_x512": small["2^11_x512"]["gf_elems_per_s"], "gf_elems_per_s_2^20": sweep["2^20"]["fwd"]["gf_elems_per_s"]}
        line["msm"] = {"msms_per_s_2^11_x4608": world * 4608 / (ms_msm * 1e-3), "ms_4608": ms_msm, "replicas": world}
        ms20 = sweep["2^20"]["fwd"]["ms"]
        ach = 64.0 * (1 << 20) / (ms20 * 1e-3) / 1e9  # per GPU
        # the two launches of the 2^20 transform one by one: HIP events recorded around each pass on the library's stream
        # (profiles/r04_ntt_kernel_stats.txt holds the rocprofv3 --kernel-trace --stats durations of the same transforms)
        ctx.profile_reset()
        ctx.profile(True)
        ntt_microbench(ctx, 20, 1, reps=8)
        ctx.profile(False)
        pc, pr_ = ctx.profile_read("ntt_pass_columns"), ctx.profile_read("ntt_pass_rows")
        per_pass = {"columns_us": 1e3 * pc[0] / max(pc[1], 1), "rows_us": 1e3 * pr_[0] / max(pr_[1], 1)}
        ctx.profile_reset()
        tr20 = pmc.get("ntt", {}).get("ntt_2^20")
        q20 = sweep["2^20"]["fwd_queue16"]["ms_per_transform"]
        line["roofline_ntt"] = {"kernel": "ntt_wavel_column_kernel + ntt_wavel_kernel (N = 2^20 = 2^10 x 2^10, two launches)", "bound": "hbm",
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                "per_pass_us": per_pass, "ms_lone": ms20, "ms_in_a_queue_of_16": q20,
                                "frac_in_a_queue_of_16": 64.0 * (1 << 20) / (q20 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": tr20, "traffic_over_algorithmic": tr20 / (64.0 * (1 << 20)) if tr20 else None, "traffic_source": pmc_src,
                                "traffic_note": "includes 80 N bytes of inter-pass twiddles read from the table in usage order (one "
                                                "multiplication per element instead of two, a deliberate bytes-for-instructions trade; "
                                                "plonk_ntt_set_table_budget(0) gives 2.07 x the algorithmic 64 N instead of 3.3 x and a 6 % slower transform)",
                                "note": "ALU-bound on the 254-bit multiplication: ~9.5 N multiplications (two passes + inter-pass "
                                        "twiddles) at ~150-170 G/s chip-wide bound the transform near 10 % of HBM peak (DESIGN.md 4.1)"}
        if "roofline" in line:  # the kernel north_star puts a number on, inside the block the driver's record keeps
            line["roofline"]["secondary"] = {k: line["roofline_ntt"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "per_pass_us",
                                                                                     "ms_lone", "ms_in_a_queue_of_16", "frac_in_a_queue_of_16", "traffic",
                                                                                     "traffic_over_algorithmic", "traffic_source")}
        line["config"]["ntt_2^20_ms"] = ms20
        line["config"]["ntt_2^16_ms"] = sweep["2^16"]["fwd"]["ms"]
        line["config"]["ntt_2^24_ms"] = sweep["2^24"]["fwd"]["ms"]
        line["config"]["ntt_2^11_x2048_gf_elems_s"] = small["2^11_x2048"]["gf_elems_per_s"]
        line["config"]["bls12_381_ntt_parity"] = BLS_PIN_NOTE
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dt, oproof, prim = cpu_baseline()
        line["cpu_baseline"] = {
            "value": 1.0 / dt,
            "unit": "proofs/s",
            "cores": 1,
            "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": "1 full proof of the same group_order=2^11 circuit by oracle/plonk_prover.py (pure Python), %.1f s; "
                      "primitives: best of 3 each" % dt,
            "primitives": prim,
        }
        # the GPU proof of the same witness must be bit-identical to the oracle's
        got = BatchProver.decode(gathered[0]).flatten()
        want = oproof.flatten()
        same = all(
            ((got[k][0].n, got[k][1].n) if isinstance(got[k], tuple) else got[k].n) == want[k] for k in want
        )
        line["cpu_baseline"]["gpu_proof_bit_identical"] = bool(same)
        if "latency" in line:  # north_star's target is a latency ratio: the reference-CPU proof time over one GPU proof
            for k in ("api_prover_with_asserts", "api_prover", "batch_prover_b1"):
                line["latency"]["speedup_vs_cpu_proof_" + k] = dt / (line["latency"][k]["median_ms"] * 1e-3)
        if "ntt" in line:
            line["cpu_baseline"]["gpu_speedup_fft_2^11"] = prim["fft_2^11_ms"] / (line["ntt"]["ms_2^11_x512"] / 512)
            line["cpu_baseline"]["gpu_speedup_ec_lincomb_2^11"] = prim["ec_lincomb_2^11_s"] * 1e3 / (line["msm"]["ms_4608"] / 4608)
            cp = prim.get("c", {})
            if "ntt_2^11_ms" in cp:  # against the compiled single-core restatement
                line["cpu_baseline"]["gpu_speedup_vs_c_ntt_2^11"] = cp["ntt_2^11_ms"] / (line["ntt"]["ms_2^11_x512"] / 512)
                line["cpu_baseline"]["gpu_speedup_vs_c_ntt_2^20"] = cp["ntt_2^20_ms"] / line["ntt"]["ms_2^20"]
                line["cpu_baseline"]["gpu_speedup_vs_c_g1_lincomb_2^11"] = cp["g1_lincomb_2^11_ms"] / (line["msm"]["ms_4608"] / 4608)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
