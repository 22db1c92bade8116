#!/usr/bin/env python3
"""bench.py — proofs/sec of the plonkathon prover hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1 via torch.distributed.run, one rank/GPU)

A "step" is one pass of the hot path over one batch: `--batch` independent PLONK proofs per GPU of
the BASELINE configs[1] workload (group_order = 2^11, the powers-of-tau SRS slice, synthetic witness
— a 2047-gate squaring chain + one public input, one distinct witness per proof).  Proofs are independent,
so N GPUs shard by proof index with no data-path collective; the final (9 G1 + 6 Fr = 768 B) results
are gathered with one RCCL all_gather ("scaling": "weak").  Inputs (circuit polynomials, the MSM lookup
table of the SRS, witness columns) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line: the contract fields, plus
  "roofline"      for the dominant kernel of the timed region (msm_lookup; msm_accumulate if no table fits),
                  durations from HIP events recorded on the library's stream inside the timed region;
  "roofline_ntt"  the standalone Fr NTT at 2^20 (BASELINE configs[3]) against the HBM roofline;
  "ntt", "msm"    NTT GF-elems/s at 2^11 (batched) and 2^20; MSMs/s at 2^11 (512 x 9 commitments per call);
                  N replicas for N GPUs;
  "cpu_baseline"  the oracle (pure-Python port of the reference path) timed on this box, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
# chip-wide rate of the MSM loop's unit of work — the lazy mixed addition on register-resident operands —
# measured by tools/ubench (profiles/r01_k_ubench.json: g1_lazy_madd_Gops; fq_lazy_mul_Gops = 167)
G1_MADD_CEILING_G = 13.5
MSM_WINDOW_BITS = 10      # bucket-method default (csrc/msm.hip); 26 windows of signed 10-bit digits
GROUP_ORDER = 2048
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")


def chain_program_lines(n):
    """one public input + a squaring chain; every wire value is non-zero."""
    return ["x0 public"] + ["x%d <== x%d * x%d" % (i + 1, i, i) for i in range(n - 1)]


def witness_for(program, proof_index):
    return program.fill_variable_assignments({"x0": 3 + proof_index})


def proof_bytes(proof):
    out = b""
    for k, v in proof.flatten().items():
        if isinstance(v, tuple):
            out += v[0].n.to_bytes(32, "big") + v[1].n.to_bytes(32, "big")
        else:
            out += v.n.to_bytes(32, "big")
    return out


def cpu_baseline():
    """One full proof of the same workload by the oracle on one host core (the reference is
    single-threaded pure Python)."""
    from oracle.circuit import Program as OProgram
    from oracle.plonk_prover import Prover as OProver
    from oracle.srs import Setup as OSetup

    prog = OProgram(chain_program_lines(GROUP_ORDER), GROUP_ORDER)
    wit = prog.fill_variable_assignments({"x0": 3})
    prover = OProver(OSetup.from_file(PTAU), prog)
    t0 = time.perf_counter()
    proof = prover.prove(dict(wit))
    dt = time.perf_counter() - t0
    return dt, proof


def ntt_microbench(ctx, log_n, batch, reps=5):
    from plonkathon_amd._lib import check

    n = 1 << log_n
    import random

    rng = random.Random(log_n)
    # device-side fill: upload one random block and replicate it (content does not affect timing)
    per = min(n * batch, 4096)
    src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(per)])
    buf = ctx.alloc(n * batch)
    for off in range(0, n * batch, per):
        check(ctx.L.plonk_mem_d2d(ctx.handle, buf.at(off), src.ptr, 32 * min(per, n * batch - off)))
    out = ctx.alloc(n * batch)
    check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 0, batch))  # warm: tables + scratch
    ctx.sync()
    best = None
    for _ in range(reps):
        ctx.timer_start()
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 0, batch))
        ms = ctx.timer_stop_ms()
        best = ms if best is None or ms < best else best
    return best


def msm_microbench(ctx, setup, batch, reps=3):
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
    bases = setup.device_bases(ctx)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=512, help="proofs per GPU per step")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams per GPU: the batch is split into this many lock-step sub-batches that overlap each other")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--lookup-budget-gb", type=float, default=0.0,
                    help="HBM budget for the MSM lookup table (0 = library default: 55 %% of free memory, at most 160 GB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    from plonkathon_amd import BatchProver, Context, Program, Setup, set_context
    from plonkathon_amd import distributed as D

    dist = D.init_from_env(args.dist_backend) if world > 1 else None
    ctx = Context(local_rank)
    set_context(ctx)
    if args.lookup_budget_gb:
        ctx.msm_lookup(0, 0, int(args.lookup_budget_gb * 1e9))
    setup = Setup.from_file(PTAU)
    program = Program(chain_program_lines(GROUP_ORDER), GROUP_ORDER)
    B = args.batch
    S = max(1, min(args.streams, B))
    total = B * world  # weak scaling: every GPU proves B proofs per step
    mine = D.shard_indices(total, rank, world)
    # S lock-step sub-batches on S HIP streams of the same GPU: while one sub-batch sits in a latency-bound
    # kernel (transcript, field inversions), the others keep the ALUs busy with MSM / NTT work
    ctxs = [ctx] + [Context(local_rank) for _ in range(S - 1)]
    provers = [BatchProver(setup, program, c) for c in ctxs]
    # synthetic witnesses, one per GLOBAL proof index (all distinct: identical proofs would turn the MSM's table
    # look-ups into cache hits); staged in HBM before the timed region
    parts = [mine[k::S] for k in range(S)]
    for pr, part in zip(provers, parts):
        pr.upload([witness_for(program, idx) for idx in part])

    def step():
        for pr in provers:
            pr.run()                   # five rounds + transcript: one stream of kernel launches each
        blobs = [pr.download_raw() for pr in provers]   # sync + 768 B per proof back to the host
        if S == 1:
            return blobs[0][0], bytes(blobs[0][1])
        out, status = [None] * len(mine), [0] * len(mine)
        for k, (raw, st) in enumerate(blobs):
            for j in range(len(parts[k])):
                out[k + S * j] = raw[768 * j : 768 * (j + 1)]
                status[k + S * j] = st[j]
        return b"".join(out), bytes(status)

    def barrier():
        for c in ctxs:
            c.sync()
        if dist is not None:
            if dist.get_backend() == "nccl":
                import torch

                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        proofs = step()
    ctx.profile_reset()
    ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proofs = step()
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dist)
    ctx.profile(False)
    assert not any(proofs[1]), "a proof in the batch reported a failure status"
    # the only collective on the path: all_gather of the finished proofs (768 B each) over RCCL/xGMI
    gathered = D.gather_proofs(proofs[0], total, dist)
    n_results = len(gathered)

    # the dominant kernel: the lookup MSM when the table fits in HBM (default), else the bucket method's accumulate
    lookup_bits = setup.device_bases(ctx).lookup_bits
    msm_kernel = "msm_lookup" if lookup_bits else "msm_accumulate"
    msm_ms, msm_launches, msm_bytes = ctx.profile_read(msm_kernel)  # stream 0's launches
    total_proofs = args.steps * B * world
    line = {
        "metric": "proofs/sec at group_order=2^11 (PLONK prover hot path: NTT + quotient + KZG MSM)",
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
            "workload": "configs[1]: group_order=2^11, powersOfTau28_hez_final_11 SRS slice, synthetic squaring-chain witness",
            "proofs_per_gpu_per_step": B,
            "prover": "BatchProver (lock-step, GPU-resident transcript)",
            "streams_per_gpu": S,
            "results_gathered": n_results,
            "parallelism": "proof-sharded x%d" % world,
        },
    }
    def pmc_traffic(kernel, run):
        """HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)."""
        path = os.path.join(REPO, "profiles", "r01_pmc_summary.json")
        if not os.path.exists(path):
            return None, None
        ks = [k for k in json.load(open(path))["kernels"] if k["kernel"] == kernel and k["run"] == run]
        n = sum(k["launches"] for k in ks)
        if not n:
            return None, None
        return sum(k["traffic_bytes"] * k["launches"] for k in ks) / n, "profiles/r01_pmc_summary.json"

    if msm_launches:
        avg_s = msm_ms * 1e-3 / msm_launches
        achieved = (msm_bytes / msm_launches) / avg_s / 1e9
        line["roofline"] = {
            "kernel": msm_kernel + "_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(msm_kernel + "_kernel", "bench")[0] if B // S == 512 else None,
            "traffic_source": "profiles/r01_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                              "`bench.py --batch 512 --streams 1` (tools/pmc_collect.sh), 2*FETCH+WRITE, launch-weighted mean",
            "launches": msm_launches,
            "avg_launch_us": avg_s * 1e6,
            "note": "algorithmic bytes = 96*N+64 per MSM (SURVEY.md 8(d)); the kernel is integer-ALU bound (DESIGN.md 3/4.2), "
                    "see `alu`; with the lookup table every addition also reads 64 table bytes, i.e. `traffic` is the real demand",
            "msm_method": ("lookup table, %d-bit windows" % lookup_bits) if lookup_bits else "bucket method, %d-bit windows" % MSM_WINDOW_BITS,
        }
        if lookup_bits:  # the lookup method's own algorithmic bytes: one 64-byte table entry per addition + the scalars
            windows_l = (255 + lookup_bits - 1) // lookup_bits
            per_msm = GROUP_ORDER * windows_l * 64.0 + 32.0 * GROUP_ORDER + 64.0
            n_msm_l = msm_bytes / (96.0 * GROUP_ORDER + 64.0)
            line["roofline"]["method_bytes_per_msm"] = per_msm
            line["roofline"]["method_GBps"] = per_msm * n_msm_l / (msm_ms * 1e-3) / 1e9
            line["roofline"]["method_frac_of_peak"] = line["roofline"]["method_GBps"] / HBM_PEAK_GBS
        if line["roofline"]["traffic"]:  # what the kernel really asks of HBM (table look-ups), per the PMC passes
            line["roofline"]["traffic_GBps"] = line["roofline"]["traffic"] / avg_s / 1e9
            line["roofline"]["traffic_frac_of_peak"] = line["roofline"]["traffic_GBps"] / HBM_PEAK_GBS
        # the honest ceiling: W*N mixed additions per MSM against the rate of a bare mixed-addition loop
        n_msm = msm_bytes / (96.0 * GROUP_ORDER + 64.0)
        wbits = lookup_bits or MSM_WINDOW_BITS
        windows = (255 + wbits - 1) // wbits
        gmadd = n_msm * windows * GROUP_ORDER / (msm_ms * 1e-3) / 1e9
        line["roofline"]["alu"] = {"achieved_g1_gmadd_per_s": gmadd, "ceiling_g1_gmadd_per_s": G1_MADD_CEILING_G,
                                   "frac": gmadd / G1_MADD_CEILING_G,
                                   "note": "%d mixed additions per MSM (8 Fq mul + 2 sqr + 8 add/sub each); ceiling = the same "
                                           "addition in a register-only loop (tools/ubench), i.e. the kernel adds no overhead "
                                           "beyond the arithmetic itself" % (windows * GROUP_ORDER)}
    if not args.no_microbench:
        # standalone NTT and MSM rates; with N GPUs every rank runs a replica and the whole-job
        # rate is N x (work of one replica) / (time of the slowest rank)
        ms11 = D.max_over_ranks(ntt_microbench(ctx, 11, 512), dist)
        ms20 = D.max_over_ranks(ntt_microbench(ctx, 20, 1), dist)
        ms_msm = D.max_over_ranks(msm_microbench(ctx, setup, 4608), dist)
        line["ntt"] = {
            "gf_elems_per_s_2^11_x512": world * 512 * 2048 / (ms11 * 1e-3),
            "gf_elems_per_s_2^20": world * (1 << 20) / (ms20 * 1e-3),
            "ms_2^11_x512": ms11,
            "ms_2^20": ms20,
            "replicas": world,
        }
        line["msm"] = {"msms_per_s_2^11_x4608": world * 4608 / (ms_msm * 1e-3), "ms_4608": ms_msm, "replicas": world}
        ach = 64.0 * (1 << 20) / (ms20 * 1e-3) / 1e9  # per GPU
        line["roofline_ntt"] = {"kernel": "ntt_pass_kernel (2 passes, N=2^20)", "bound": "hbm", "achieved": ach,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dt, oproof = cpu_baseline()
        line["cpu_baseline"] = {
            "value": 1.0 / dt,
            "unit": "proofs/s",
            "cores": 1,
            "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": "1 full proof of the same group_order=2^11 circuit by oracle/plonk_prover.py (pure Python), %.1f s" % dt,
        }
        # the GPU proof of the same witness must be bit-identical to the oracle's
        got = BatchProver.decode(gathered[0]).flatten()
        want = oproof.flatten()
        same = all(
            ((got[k][0].n, got[k][1].n) if isinstance(got[k], tuple) else got[k].n) == want[k] for k in want
        )
        line["cpu_baseline"]["gpu_proof_bit_identical"] = bool(same)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
