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
the defaults a step is 20 x 512 = 10240 proofs per GPU (~0.24 s), so that the driver's 20 timed steps last ~5 s.
Proofs are independent, so N GPUs shard by proof index with no data-path collective; every step ends with the one
collective of the path, an all-gather of the finished proofs (768 B each) over RCCL ("scaling": "weak").  Inputs
(circuit polynomials, the MSM comb table of the SRS, witness columns) are resident in HBM before the timed region.

OUTPUT.  stdout carries ONE JSON line of under 4 KB (asserted): the contract fields, `config` (workload + a dozen
scalars), `roofline` (the dominant kernel, msm_comb_kernel, against the HBM roofline — durations from HIP events on the
library's own streams, traffic from this round's committed PMC passes, the ALU view from this round's tools/ubench run
and SQ counter passes — with `secondary` = the standalone 2^20 transform north_star puts a number on) and `cpu_baseline`
(the oracle on one host core; the GPU proof of the same witness compared byte for byte).  EVERYTHING ELSE the run
measures — BASELINE configs[2] (Poseidon) and configs[3] (2^16 .. 2^24 in both fields), the fallbacks (smaller tables,
the bucket method, ec_lincomb on arbitrary bases), fresh uploads inside the timed region, single-proof latencies, host
costs, clocks, per-rank figures, notes — goes to `bench_detail.json` (`--detail PATH`) and, as one line, to stderr.
The legs live in tools/bench_legs.py.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import bench_legs as legs  # noqa: E402  (tools/bench_legs.py)
from bench_legs import HBM_PEAK_GBS, NOMINAL_SCLK_MHZ, R_MOD, ClockSampler  # noqa: E402,F401  (ClockSampler: re-exported for the tests)

MSM_WINDOW_BITS = 10      # bucket-method default (csrc/msm.hip); 26 windows of signed 10-bit digits
GROUP_ORDER = 2048
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")
def table_adds(info, n):
    """mixed additions of one MSM of n scalars on the table `info` describes (a comb with top tables: its virtual scalars too)"""
    a, g = info["additions_per_base"], info.get("top_group", 0)
    return a * (n + (-(-(-(-n // g)) // a) if g else 0))


# opt-in budget for the MSM table: the comb of 21 teeth with top tables over 2^11 bases is 157.6 GB + 17.2 GB of build staging (12.15
# additions per base); 100 buys the plain comb of 20 teeth (68.7 GB, 13 additions: rounds 5's headline), the library's own default
# (1/16 of the device's memory) the one of 17 teeth (8.6 GB, 15)
DEFAULT_TABLE_GB = 180.0
LINE_LIMIT = 4096         # bytes of the stdout line (the driver's record keeps the last 8 KB of stdout)


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


EXIT_PEER_LOST = 76  # this rank is healthy but another one stopped answering (a collective's deadline, a closed socket)
EXIT_INJECTED = 17   # --inject-fault (tests)


def _tail(path, n=12):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            f.seek(max(0, f.tell() - 16384))
            return [l for l in f.read().decode("utf-8", "replace").splitlines() if l.strip()][-n:]
    except OSError:
        return []


def spawn_ranks(args):
    """`--gpus N` without a launcher: one child process per GPU, rank 0's JSON line is ours.  EVERY child is watched: the first one
    that exits non-zero ends the job — the others are stopped, and the report names the rank that failed first (a rank that only
    lost a peer exits with EXIT_PEER_LOST and is not blamed while another rank died of its own), its exit code or signal and the tail
    of its stderr.  A rank that hangs is ended by the deadlines inside the ranks (--comm-timeout), not here."""
    import ctypes
    import signal
    import socket
    import tempfile

    from plonkathon_amd import _lib

    n_dev = ctypes.c_int(0)
    _lib.check(_lib.lib().plonk_device_count(ctypes.byref(n_dev)))
    if args.dist_backend == "rccl" and n_dev.value < args.gpus:
        sys.exit("bench.py: --gpus %d but only %d HIP device(s) are visible; refusing to run fewer ranks than asked "
                 "(use --dist-backend sockets to share a GPU between ranks for a functional test)" % (args.gpus, n_dev.value))
    with socket.socket() as s:  # a free port pair for the rendezvous
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="plonk_bench_ranks_")
    procs, outs, errs = [], [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PLONK_RDZV_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PLONK_JOB_ID=os.environ.get("PLONK_JOB_ID", "bench-%d-%d" % (os.getpid(), port)))
        outs.append(os.path.join(logdir, "rank%d.out" % r))
        errs.append(os.path.join(logdir, "rank%d.err" % r))
        with open(outs[r], "wb") as fo, open(errs[r], "wb") as fe:
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=fo, stderr=fe))
    order = []  # ranks in the order their exit was seen
    first_bad_at = None
    while len(order) < len(procs):
        for r, p in enumerate(procs):
            if r not in order and p.poll() is not None:
                order.append(r)
                if p.returncode != 0 and first_bad_at is None:
                    first_bad_at = time.monotonic()
        # after the first failure the others get two seconds to notice by themselves (their own message is worth more than ours)
        if first_bad_at is not None and time.monotonic() - first_bad_at > 2.0:
            break
        time.sleep(0.02)
    stopped = [r for r in range(len(procs)) if procs[r].poll() is None]
    for r in stopped:
        procs[r].terminate()
    t_kill = time.monotonic() + 5.0
    for r in stopped:
        try:
            procs[r].wait(timeout=max(0.1, t_kill - time.monotonic()))
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    rcs = [p.returncode for p in procs]
    with open(errs[0], "rb") as f:  # rank 0's stderr is ours (the detail record travels on it)
        sys.stderr.write(f.read().decode("utf-8", "replace"))
    with open(outs[0], "rb") as f:
        out = f.read().decode("utf-8", "replace")
    bad = [r for r in order if rcs[r] != 0]
    if not bad:
        sys.stdout.write(out)
        sys.stdout.flush()
        import shutil

        shutil.rmtree(logdir, ignore_errors=True)
        return

    def how(rc):
        if rc < 0:
            try:
                return "was killed by %s" % signal.Signals(-rc).name
            except ValueError:
                return "was killed by signal %d" % -rc
        return "exited with code %d" % rc + (" (it lost a peer)" if rc == EXIT_PEER_LOST else "")

    own = [r for r in bad if rcs[r] != EXIT_PEER_LOST]
    culprit = (own or bad)[0]
    if own or not stopped:
        sys.stderr.write("bench.py: rank %d of %d %s first; the job was stopped.\n" % (culprit, args.gpus, how(rcs[culprit])))
    else:  # nobody died of its own: the ranks that gave up were waiting for one that never answered and was still running
        sys.stderr.write("bench.py: rank %s of %d did not exit and had to be stopped (stuck?); rank %d %s waiting for it; the job was stopped.\n"
                         % (", ".join(map(str, stopped)), args.gpus, culprit, how(rcs[culprit])))
    for r in range(args.gpus):
        state = "stopped by the launcher while still running" if r in stopped else how(rcs[r])
        last = _tail(errs[r], 1)
        sys.stderr.write("bench.py:   rank %d %s%s\n" % (r, state, (" | last stderr line: " + last[0][:300]) if last and r != culprit else ""))
    if culprit != 0:
        sys.stderr.write("bench.py: stderr tail of rank %d:\n" % culprit)
        for l in _tail(errs[culprit]):
            sys.stderr.write("bench.py:     " + l[:400] + "\n")
    sys.stderr.write("bench.py: per-rank logs kept in %s\n" % logdir)
    sys.exit(1)


class Watchdog:
    """Host-side deadline for the multi-rank phases of this process: a daemon thread that ends the process — naming
    the rank and what it was waiting in — when an armed phase outlives its deadline.  The transports have deadlines of their own
    (RCCL: plonk_comm_set_timeout -> PLONK_ERR_TIMEOUT; sockets: PeerLost); this is the backstop for a wait neither of them sees
    (a GPU that stops answering, a rank stuck before its first collective)."""

    def __init__(self, rank, world):
        import threading

        self.rank, self.world, self.deadline, self.label = rank, world, None, ""
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def arm(self, label, seconds):
        self.label, self.deadline = label, (time.monotonic() + seconds if seconds > 0 else None)

    def disarm(self):
        self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.25)
            d = self.deadline
            if d is not None and time.monotonic() > d:
                sys.stderr.write("bench.py[rank %d of %d]: '%s' outlived its deadline: another rank is dead or stuck, or this rank's GPU "
                                 "stopped answering - giving up\n" % (self.rank, self.world, self.label))
                sys.stderr.flush()
                os._exit(EXIT_PEER_LOST)


def _round(x, digits=6):
    """Floats of the stdout line to 6 significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _round(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, digits) for v in x]
    return x


def emit(line, detail, detail_path):
    """ONE line on stdout, under LINE_LIMIT bytes; the detail beside it and on stderr."""
    line = _round(line)
    text = json.dumps(line, separators=(",", ":"))
    for optional in (("per_rank", "proofs_per_s"), ("roofline", "secondary", "valu"), ("roofline", "valu"), ("roofline", "step_valu_source"), ("roofline", "valu_source"),
                     ("roofline", "traffic_source"), ("per_rank",), ("roofline", "secondary")):
        if len(text) < LINE_LIMIT:
            break
        d = line  # shed optional blocks (they stay in the detail file) rather than print a line the driver cannot keep
        for k in optional[:-1]:
            d = d.get(k, {})
        d.pop(optional[-1], None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, "bench line is %d bytes" % len(text)
    detail = dict(detail, line=line)
    try:
        with open(detail_path, "w") as f:
            json.dump(detail, f, indent=1)
    except OSError as exc:
        sys.stderr.write("bench.py: could not write %s: %r\n" % (detail_path, exc))
    sys.stderr.write("bench_detail: " + json.dumps(detail) + "\n")
    sys.stderr.flush()
    print(text, flush=True)


def preflight(args, ctx, comm, rank, world, local_rank, setup, comm_init_s, step_s, gather_ms, device_gather):
    """`--preflight`: what every rank sees, gathered by rank 0 into ONE JSON line — the questions a failed or slow N-GPU run raises
    first (is every rank on its own device, how much HBM is free, can the devices map each other's memory, which RCCL, how long did
    the communicator and the MSM table take, what does one all-gather cost)."""
    import ctypes

    free, total = ctx.mem_info()
    n_dev = ctypes.c_int(0)
    ctx.L.plonk_device_count(ctypes.byref(n_dev))
    row = (ctypes.c_int * max(n_dev.value, 1))()
    peer = list(row) if ctx.L.plonk_device_peer_access(ctx.device, row, len(row)) == 0 else None
    info = setup.device_bases(ctx).lookup_info()
    rec = {"rank": rank, "local_rank": local_rank, "device": ctx.device, "devices_visible": n_dev.value, "name": ctx.name(), "pid": os.getpid(),
           "hbm_free_gb": round(free / 1e9, 2), "hbm_total_gb": round(total / 1e9, 2), "peer_access": peer,
           "comm_init_s": round(comm_init_s, 3), "first_step_s": round(step_s[0], 3), "next_step_ms": round(1e3 * step_s[-1], 2),
           "msm_table": {"layout": info["layout"], "bits": info["bits"], "gb": round(info["bytes"] / 1e9, 2), "build_s": round(info["build_s"], 3)},
           "transport": comm.kind if comm is not None else "none (single rank)"}
    if comm is not None and comm.kind == "rccl":
        ri = comm.info()
        rec.update({"rccl_path": ri["path"], "rccl_version": ri["version"]})
    if comm is not None and gather_ms[2]:
        rec["allgather_us"] = round(1e3 * gather_ms[0] / gather_ms[2], 1)
        rec["allgather_measured_by"] = "HIP events around ncclAllGather" if device_gather else "host clock around the exchange"
    if comm is not None:
        blob = json.dumps(rec).encode()
        assert len(blob) <= 4096
        rows = [json.loads(b.rstrip(b"\0").decode()) for b in comm.all_gather(blob + bytes(4096 - len(blob)))]
    else:
        rows = [rec]
    if rank == 0:
        devs = [r["device"] for r in rows]
        report = {"preflight": True, "n_gpus": world, "ranks_in_communicator": comm.world if comm is not None else 1,
                  "distinct_devices": len(set(devs)), "all_peers_reachable": all(all(r["peer_access"] or [0]) for r in rows), "ranks": rows}
        sys.stderr.write("bench_preflight: " + json.dumps(report, indent=1) + "\n")
        print(json.dumps(_round(report), separators=(",", ":")), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


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
    ap.add_argument("--streams", type=int, default=0, help="HIP streams per GPU (0 = one per lock-step batch of a step: --batches-per-step): the lock-step batches of a step are dealt round-robin to this many contexts, so one batch's latency-bound kernels (transcript, inversions, scans) overlap another's MSMs")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "sockets"],
                    help="transport of the final gather for N > 1: rccl = RCCL over xGMI through the C-ABI (default); sockets = TCP, lets ranks share one GPU")
    ap.add_argument("--lookup-budget-gb", type=float, default=DEFAULT_TABLE_GB,
                    help="HBM budget for the MSM table (the library's own default is 1/16 of the device's memory; 100 = the 20-tooth comb of 2^11 bases, 68.7 GB)")
    ap.add_argument("--lookup-bits", type=int, default=0,
                    help="teeth of the comb table (0 = the largest the budget affords: 20 for 2^11 bases under the default budget); A/B runs")
    ap.add_argument("--force-comm", action="store_true",
                    help="with --gpus 1: still create a ONE-rank RCCL communicator and run the gather (plonk_gather_proofs_device), the "
                         "max over ranks and the barrier inside the timed region — the code path of an N-GPU run, exercised on one GPU")
    ap.add_argument("--comm-timeout", type=float, default=120.0,
                    help="N > 1: seconds a collective of a step may wait for the other ranks before this rank gives up (RCCL: plonk_comm_set_timeout "
                         "-> the communicator is aborted, PLONK_ERR_TIMEOUT; sockets: PeerLost); a host-side watchdog at 1.5 x + 30 s backs it up")
    ap.add_argument("--init-timeout", type=float, default=300.0, help="N > 1: seconds for the rendezvous and ncclCommInitRank")
    ap.add_argument("--preflight", action="store_true",
                    help="set everything up (communicator, MSM table, one step with its gather), print ONE JSON line with a record per rank - device, "
                         "free HBM, hipDeviceCanAccessPeer row, RCCL path / version, communicator and table build seconds, one all-gather in us - and exit")
    ap.add_argument("--inject-fault", default="", metavar="RANK:STEP:MODE",
                    help="tests: rank RANK fails in timed step STEP (0-based) between enqueueing its batches and the gather; MODE = exit (os._exit), "
                         "kill (SIGKILL) or hang (sleeps for ever)")
    ap.add_argument("--no-lookup", action="store_true", help="bucket-method MSM only")
    ap.add_argument("--host-gather", action="store_true", help="N > 1 with RCCL: gather through host buffers (plonk_gather_results) instead of straight from the provers' device buffers (plonk_gather_proofs_device)")
    ap.add_argument("--dump-proofs", default="", help="rank 0 writes the last step's gathered proofs (768 bytes each, global order) to this file")
    ap.add_argument("--lagrange-commits", action="store_true", help="commit rounds 1-2 from Lagrange values over the Lagrange-basis SRS (a second table)")
    ap.add_argument("--msm-groups", type=int, default=-1, help="plonk_msm_configure groups: workgroups per MSM (0 = library default; -1 = 1 with eight or more streams, else the library default)")
    ap.add_argument("--ntt-kind", type=int, default=0, help="plonk_ntt_select_kernel: 0 auto, 1 / 4 the LDS kernel (radix-2 stages; A/B), 5 wave kernels wherever they apply, 6 / 7 wave kernels without / with the latency forms")
    ap.add_argument("--log-n", type=int, default=11, help="log2(group_order); 11 = the BASELINE workload, smaller values are for functional tests only")
    ap.add_argument("--detail", default=os.path.join(REPO, "bench_detail.json"), help="where the full record goes (the stdout line stays under 4 KB)")
    ap.add_argument("--verify-samples", type=int, default=4, help="random proofs of the last step put under the pairing check, untimed (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-fallbacks", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs[2] (Poseidon)")
    ap.add_argument("--no-latency", action="store_true")
    args = ap.parse_args()

    # RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails otherwise); the GPU boxes export this
    # already — set before the HIP runtime is loaded in case a launcher scrubbed the environment
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    # streams share queues (and serialise); more queues than the device schedules at once cost every latency-bound path
    # (profiles/r04_j_streams_hw_queues.jsonl: 20 streams on 20 queues +4-5 % over 4 on 4; from 24 queues up the one-stream legs lose)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues or min(20, max(4, args.streams or args.batches_per_step))))

    from plonkathon_amd import BatchProver, Context, Program, Setup, set_context
    from plonkathon_amd import distributed as D

    ctx = Context(local_rank)
    set_context(ctx)
    multi = world > 1 or args.force_comm
    wd = Watchdog(rank, world) if multi else None
    guard = 1.5 * args.comm_timeout + 30.0  # the host-side backstop fires after the transport's own deadline had its chance

    def phase(msg):  # one stderr line per phase of a multi-rank run: what a rank was doing when the job stopped
        if world > 1:
            sys.stderr.write("bench.py[rank %d of %d, device %d, pid %d]: %s\n" % (rank, world, ctx.device, os.getpid(), msg))
            sys.stderr.flush()

    fault = None
    if args.inject_fault:
        fr, fk, fmode = args.inject_fault.split(":")
        if int(fr) == rank:
            fault = (int(fk), fmode)
    phase("context up; creating the communicator (%s)" % args.dist_backend)
    t_comm = time.perf_counter()
    if wd:
        wd.arm("communicator set-up (rendezvous + ncclCommInitRank)", args.init_timeout + 30.0)
    # librccl announces itself on C stdout ("RCCL version : ..", five lines, flushed whenever libc pleases — after this script's JSON
    # line when stdout is a pipe): while the communicator is created, file descriptor 1 points at stderr, and libc's buffer is flushed
    # before it is restored, so that stdout carries the one JSON line and nothing else
    import ctypes

    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        comm = D.init_from_env(ctx, args.dist_backend, timeout=args.init_timeout) if world > 1 else None
        if comm is None and args.force_comm:
            comm = D.RcclComm(ctx, 0, 1, args.init_timeout) if args.dist_backend == "rccl" else D.SocketComm(0, 1, args.init_timeout)
        if comm is not None and comm.kind == "rccl":
            comm.barrier()  # (the first collective: whatever the library prints lazily, it prints now)
            ctx.sync()
        if comm is not None:
            comm.set_timeout(args.comm_timeout)  # from here on the ranks move in step: a collective that waits longer has lost a rank
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    comm_init_s = time.perf_counter() - t_comm
    if wd:
        wd.disarm()
    if comm is not None and comm.world != world:
        sys.exit("bench.py: communicator has %d ranks, expected %d" % (comm.world, world))
    phase("communicator up in %.2f s; staging witnesses" % comm_init_s)
    budget = 0 if args.no_lookup else int(args.lookup_budget_gb * 1e9)
    B, S = args.batch, args.batches_per_step
    NS = max(1, args.streams or S)
    ctxs = [ctx] + [Context(local_rank) for _ in range(NS - 1)]
    # Workgroups per MSM: the library cuts an MSM into as many workgroups as fill the chip in whole rounds ON ITS OWN (1 536 MSMs
    # into 3 072 workgroups); with eight or more streams the other streams' kernels fill a partial round, and one workgroup per
    # MSM — half the per-column trees — is 0.5 - 1 % faster (profiles/r05_u_msm_groups_ab.txt).  --msm-groups overrides.
    # The knobs are per-context state: they are set through Context.tuning() and leave with `knobs.close()` right after the timed
    # region (everything measured below it runs on the library defaults unless --msm-groups / --ntt-kind pinned them explicitly)
    import contextlib

    groups = args.msm_groups if args.msm_groups >= 0 else (1 if NS >= 8 else 0)
    knobs = contextlib.ExitStack()
    for c in ctxs:
        c.msm_lookup(1 if args.no_lookup else 0, args.lookup_bits, budget)
        if args.ntt_kind:  # explicit: for the whole run
            from plonkathon_amd._lib import check as _check

            _check(c.L.plonk_ntt_select_kernel(c.handle, args.ntt_kind))
        if args.msm_groups >= 0:  # explicit: for the whole run
            c.msm_configure(0, args.msm_groups)
        elif groups:
            knobs.enter_context(c.tuning(msm_groups=groups))
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

    timed_step = [-1]  # index of the timed step being run (-1: warm-up)

    def step():
        for pr in provers:
            pr.run()                   # five rounds + transcript: one stream of kernel launches each
        if fault is not None and fault[0] == timed_step[0]:  # --inject-fault (tests): this rank dies or stalls mid-step
            phase("INJECTED FAULT '%s' in timed step %d" % (fault[1], timed_step[0]))
            if fault[1] == "kill":
                import signal

                os.kill(os.getpid(), signal.SIGKILL)
            if fault[1] == "hang":
                while True:
                    time.sleep(3600)
            os._exit(EXIT_INJECTED)
        if device_gather:              # proofs go from the provers' device buffers into the all-gather, one host copy at the end
            gathered, status = D.gather_proofs_device(provers, B, total, comm)
            a_ms, h_ms = comm.last_gather_ms()   # HIP events around the ncclAllGather and the copy to the host
            gather_ms[0] += a_ms
            gather_ms[1] += h_ms
            gather_ms[2] += 1
            return gathered.parts[rank], status, gathered
        raw = [pr.download_raw() for pr in provers]   # sync + 768 B per proof back to the host
        local = b"".join(b[0] for b in raw)
        status = b"".join(b[1] for b in raw)
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

    if args.preflight:
        args.warmup = max(2, args.warmup)
    step_s = []
    for i in range(args.warmup):
        if wd:  # (the first step builds the MSM table: seconds that differ from rank to rank)
            wd.arm("warm-up step %d (its gather waits for every rank; the first one also builds the MSM table)" % i, guard + (60.0 if i == 0 else 0.0))
        t_s = time.perf_counter()
        proofs = step()
        for c in ctxs:
            c.sync()
        step_s.append(time.perf_counter() - t_s)
        if i == 0:
            phase("first step done in %.2f s (MSM table: %s)" % (step_s[0], setup.device_bases(ctx).lookup_info()))
    if args.preflight:
        if wd:
            wd.arm("preflight report", guard)
        return preflight(args, ctx, comm, rank, world, local_rank, setup, comm_init_s, step_s, gather_ms, device_gather)
    for c in ctxs:
        c.profile_reset()
        c.profile(True)
    if wd:
        wd.arm("barrier before the timed region", guard)
    barrier()
    phase("warm-up done; timed region: %d steps" % args.steps)
    sampler = ClockSampler(local_rank) if rank == 0 else None  # one sampler per job: rank 0's GPU
    if sampler:
        sampler.start()
    gather_ms[:] = [0.0, 0.0, 0]
    t0 = time.perf_counter()
    for k in range(args.steps):
        timed_step[0] = k
        if wd:
            wd.arm("timed step %d of %d" % (k, args.steps), guard)
        proofs = step()
    timed_step[0] = -1
    for c in ctxs:
        c.sync()
    own_elapsed = time.perf_counter() - t0   # this rank's own clock, before it waits for the others
    if wd:
        wd.arm("barrier / max over ranks after the timed region", guard)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, comm)
    if wd:
        wd.disarm()
    phase("timed region done: %.3f s" % elapsed)
    clocks = sampler.summary() if sampler else None
    for c in ctxs:
        c.profile(False)

    def profile_sum(kernel):  # over every stream of this GPU
        ps = [c.profile_read(kernel) for c in ctxs]
        return sum(p[0] for p in ps), sum(p[1] for p in ps), sum(p[2] for p in ps)

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
    msm_kernel = {"comb": "msm_comb", "windows": "msm_lookup"}.get(info["layout"], "msm_accumulate")
    msm_ms, msm_launches, msm_bytes = profile_sum(msm_kernel)
    # The same kernel with the chip to itself: with several streams a launch's event-to-event duration includes the time
    # it shares the CUs with the other streams' kernels, so the per-launch figures of the timed region understate the
    # kernel.  A short untimed phase runs the batches of stream 0 alone (the other streams idle) and reads its events.
    knobs.close()  # the timed region's one-workgroup-per-MSM setting is for many streams: everything below runs one
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
    value = total_proofs / elapsed
    hbm_total = ctx.mem_info()[1]
    run = legs.Run(ctx=ctx, ctxs=ctxs, comm=comm, world=world, rank=rank, local_rank=local_rank, setup=setup, program=program, provers=provers,
                   parts=parts, blobs=blobs, mine=mine, B=B, S=S, NS=NS, per_gpu=per_gpu, steps=args.steps, value=value, barrier=barrier,
                   group_order=GROUP_ORDER, witness_for=witness_for)

    config = {
        "workload": "configs[1]: group_order=2^%d, powersOfTau28_hez_final_11 SRS slice, synthetic squaring-chain witness, one distinct witness per proof" % args.log_n,
        "proofs_per_gpu_per_step": per_gpu,
        "lockstep_batch": B,
        "batches_per_step": S,
        "streams_per_gpu": NS,
        "parallelism": "proof-sharded x%d" % world,
        "ranks_in_communicator": comm.world if comm is not None else 1,
        "gather_transport": comm.kind if comm is not None else "none (single rank)",
        "msm_method": ("comb table, %d teeth, %d columns + a joint table per %d bases for the top %d bits: %.2f additions per base"
                       % (lookup_bits, info["additions_per_base"], info["top_group"], info["top_bits"], table_adds(info, GROUP_ORDER) / GROUP_ORDER)
                       if info["layout"] == "comb" and info.get("top_group") else
                       "comb table, %d teeth: %d additions per base" % (lookup_bits, info["additions_per_base"]) if info["layout"] == "comb" else
                       "window table, %d-bit windows: %d additions per base" % (lookup_bits, info["additions_per_base"]) if lookup_bits else
                       "bucket method, %d-bit windows" % MSM_WINDOW_BITS),
        "msm_table_bytes": info["bytes"],
        "msm_table_fraction_of_hbm": info["bytes"] / hbm_total,
        "timed_region_s": elapsed,
    }
    line = {
        "metric": "proofs/sec at group_order=2^%d (PLONK prover hot path: NTT + quotient + KZG MSM)" % args.log_n,
        "value": value,
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
        "config": config,
    }
    detail = {
        "config": dict(config, **{
            "prover": "BatchProver (lock-step, GPU-resident transcript)",
            "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
            "results_gathered_per_step": n_results,
            "gather_in_timed_region": comm is not None,
            "gather_path": ("device buffers -> ncclAllGather -> host (plonk_gather_proofs_device)" if device_gather else
                            ("host buffers (plonk_gather_results / sockets)" if comm is not None else "none")),
            "msm_table_bits": lookup_bits,
            "msm_workgroups_per_msm_in_the_timed_region": groups or "library default",
            "msm_table_build_s": info["build_s"],
            "msm_table_budget_bytes": budget,
            "msm_table_shared_by": info["sharers"],
            "lagrange_commits": bool(args.lagrange_commits),
            "torch_imported": "torch" in sys.modules,  # the product and this file import no PyTorch: RCCL is reached through the C-ABI
        }),
        "host": {
            "host_upload_ms_per_proof": host_upload_ms,
            "host_upload_prepacked_ms_per_proof": host_upload_packed_ms,
            "witness_generation_ms_per_proof": 1e3 * t_gen / per_gpu,
            "witness_packing_ms_per_proof": 1e3 * t_pack / per_gpu,
            "end_to_end_proofs_per_s_from_dicts_per_gpu": per_gpu / (t_up + per_gpu * elapsed / total_proofs * world),
            "note": "host_upload: BatchProver.upload, Python witness dictionaries -> 32-byte words (V x 32 B per proof) -> HBM, wire "
                    "columns gathered on the device (timed on the first batch); prepacked: upload_values of already packed bytes "
                    "(all batches); both synchronous and outside `value` (inputs are resident before the timed region)",
        },
        "clocks": clocks,  # rank 0's GPU; None when rocm-smi prints nothing usable
    }
    if comm is not None:
        # per-rank figures, so that a scaling record explains itself: every rank's own rate (its steps over its own clock, before
        # the barrier), the seconds its MSM table took to build, and the time of the step's one collective
        import struct

        mine_stats = struct.pack("<4d", args.steps * per_gpu / own_elapsed, info["build_s"],
                                 1e3 * gather_ms[0] / max(gather_ms[2], 1), 1e3 * gather_ms[1] / max(gather_ms[2], 1))
        rows = [struct.unpack("<4d", b[:32]) for b in comm.all_gather(mine_stats)]
        rates = [r[0] for r in rows]
        detail["per_rank"] = {
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
        line["per_rank"] = {k: detail["per_rank"][k] for k in ("proofs_per_s", "proofs_per_s_min", "proofs_per_s_max", "proofs_per_s_sum",
                                                                "allgather_us_per_step_max", "allgather_fraction_of_step")}
        if comm.kind == "rccl":
            ri = comm.info()
            detail["config"].update({"rccl_path": ri["path"], "rccl_version": ri["version"], "rccl_calls_issued": ri["collectives"]})

    # HBM bytes per launch / per transform from the committed PMC passes of this round's build (rocprofv3 cannot run
    # inside this process): profiles/rNN_pmc_summary.json, written by tools/pmc_summary.py from tools/pmc_collect.sh;
    # the ALU ceilings from this round's tools/ubench run, the VALU issue counters from tools/pmc_valu.sh
    pmc, pmc_src = legs.latest_profile("pmc_summary.json")
    pmc = pmc or {"bench": {}, "ntt": {}, "factors": {}}
    ub, valu = legs.ubench_rates(), legs.valu_counters()

    if msm_launches:
        avg_s = msm_ms * 1e-3 / msm_launches
        bytes_per_launch = msm_bytes / msm_launches
        traffic = pmc["bench"].get(msm_kernel + "_kernel") if B == 512 else None
        windows = info["additions_per_base"] or (255 + MSM_WINDOW_BITS - 1) // MSM_WINDOW_BITS
        msms_per_launch = bytes_per_launch / (96.0 * GROUP_ORDER + 64.0)
        adds_per_launch = msms_per_launch * (table_adds(info, GROUP_ORDER) if info["additions_per_base"] else windows * GROUP_ORDER)
        sclk = clocks["sclk_mhz_median"] if clocks else None
        ceiling = ub["g1_lazy_madd_G"] if ub else None  # bare mixed-addition loop, millisecond burst at the nominal clock
        k_avg, k_n = (iso[0], iso[2]) if iso else (avg_s, msm_launches)  # the kernel alone on the chip when several streams ran
        gmadd = adds_per_launch / k_avg / 1e9
        roof = {
            "kernel": msm_kernel + "_kernel",
            "bound": "hbm",
            "achieved": bytes_per_launch / k_avg / 1e9,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": bytes_per_launch / k_avg / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_over_algorithmic": traffic / bytes_per_launch if traffic else None,
            "avg_launch_us": k_avg * 1e6,
            "launches": k_n,
            "frac_concurrent": bytes_per_launch / avg_s / 1e9 / HBM_PEAK_GBS,
            "g1_gmadd_per_s": gmadd,
            "alu_frac": gmadd / ceiling if ceiling else None,
            "alu_frac_at_sustained_clock": gmadd / (ceiling * sclk / NOMINAL_SCLK_MHZ) if ceiling and sclk else None,
            "alu_ceiling_gmadd_per_s": ceiling,
            "sclk_mhz": sclk,
        }
        if valu and valu.get(msm_kernel + "_kernel"):
            v = valu[msm_kernel + "_kernel"]
            roof["valu"] = {k: v.get(k) for k in ("valu_busy", "valu_insts_per_addition", "cycles_per_valu_inst")}
        # where the replayed figures come from (rocprofv3 cannot run inside this process: they are this round's committed counter
        # passes, not measurements of this run), and the WHOLE step's issue-slot account (tools/pmc_step.sh: every kernel of a
        # 20-stream step, VALU-active cycles over the step's cycles)
        roof["traffic_source"], roof["valu_source"] = pmc_src, (valu["source"] if valu else None)
        stepv, stepv_src = legs.latest_profile("step_valu.json")
        if stepv and B == 512 and info["layout"] == "comb":
            roof["step_valu_busy"] = stepv.get("step_valu_busy")
            roof["step_valu_source"] = stepv_src
        line["roofline"] = roof
        detail["roofline"] = dict(roof, **{
            "algorithmic_bytes_per_launch": bytes_per_launch, "msms_per_launch": msms_per_launch, "mixed_additions_per_msm": adds_per_launch / msms_per_launch,
            "concurrent_streams": NS, "avg_launch_us_concurrent": avg_s * 1e6, "launches_concurrent": msm_launches,
            "whole_step_g1_gmadd_per_s": (msm_bytes / (96.0 * GROUP_ORDER + 64.0)) * (adds_per_launch / msms_per_launch) / elapsed / 1e9,
            "traffic_source": pmc_src, "traffic_factors": pmc.get("factors"), "alu_source": ub["source"] if ub else None,
            "valu_counters": valu.get(msm_kernel + "_kernel") if valu else None, "valu_source": valu["source"] if valu else None,
            "traffic_GBps": traffic / k_avg / 1e9 if traffic else None,
            "note": "algorithmic bytes = 96*N+64 per MSM (SURVEY.md 8(d)); the kernel is integer-ALU bound (DESIGN.md 3/4.2): alu_frac = "
                    "mixed additions/s over the bare-loop rate of tools/ubench on this round's headers (a millisecond burst at the nominal "
                    "clock; `_at_sustained_clock` scales the ceiling by the clock sampled in the timed region); with the table every "
                    "addition also reads 64 table bytes: `traffic` (rocprofv3 --pmc FETCH_SIZE x calibrated factor + WRITE_SIZE) is the real "
                    "HBM demand; frac / avg_launch_us = the kernel with one stream active right after the timed region, frac_concurrent = "
                    "the same launches inside the %d-stream timed region (diluted by sharing the chip)" % NS})
    ntt_ms, ntt_launches, ntt_bytes = profile_sum("ntt_pass*")
    if ntt_launches:
        detail["prover_ntt"] = {"kernel": "ntt passes inside the timed prover steps", "launches": ntt_launches, "total_ms": ntt_ms,
                                "achieved_GBps": ntt_bytes / (ntt_ms * 1e-3) / 1e9, "frac_of_hbm_peak": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    if args.verify_samples and rank == 0:
        # the reference verifies what it proves (test.py:103-133): random proofs of the LAST timed step under the pairing check
        sv = legs.sampled_verify(run, gathered, total, args.verify_samples)
        detail["sampled_verify"] = sv
        config["sampled_proofs_verify"] = sv["all"]
        detail["config"]["sampled_proofs_verify"] = sv["all"]
        assert sv["all"], "a sampled proof of the last step failed verification: %s" % sv

    if not args.no_fallbacks and lookup_bits and world == 1:
        detail["fallbacks"] = legs.fallbacks(run)
    if not args.no_end_to_end and world == 1 and provers[0].variables:
        detail["end_to_end"] = legs.end_to_end(run)
    if not args.no_configs and world == 1 and args.log_n == 11:  # (smaller --log-n values are functional tests of the launch contract)
        detail["configs"] = legs.poseidon(run, poseidon_program_lines(), proof_matches_fixture)
    if not args.no_latency and world == 1:
        detail["latency"] = legs.latency(run)
    if not args.no_microbench:
        nd, roof_ntt = legs.ntt_legs(run, pmc, pmc_src, valu)
        detail.update(nd)
        if "roofline" in line:  # the kernel north_star puts a number on, inside the block the driver's record keeps
            sec = {k: roof_ntt[k] for k in ("kernel", "frac", "ms_lone", "traffic_over_algorithmic")}
            sec["per_pass_us"] = [roof_ntt["per_pass_us"]["columns_us"], roof_ntt["per_pass_us"]["rows_us"]]
            if "frac_of_alu_floor" in roof_ntt.get("alu", {}):
                sec["alu_frac"] = roof_ntt["alu"]["frac_of_alu_floor"]
            if "valu" in roof_ntt:
                sec["valu"] = roof_ntt["valu"]
            line["roofline"]["secondary"] = sec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        dt, oproof, prim = legs.cpu_baseline(PTAU, chain_program_lines(GROUP_ORDER), GROUP_ORDER)
        # the GPU proof of the same witness must be bit-identical to the oracle's
        got = BatchProver.decode(gathered[0]).flatten()
        want = oproof.flatten()
        same = all(((got[k][0].n, got[k][1].n) if isinstance(got[k], tuple) else got[k].n) == want[k] for k in want)
        line["cpu_baseline"] = {
            "value": 1.0 / dt,
            "unit": "proofs/s",
            "cores": 1,
            "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": "1 full proof of the same group_order=2^%d circuit by oracle/plonk_prover.py (pure Python), %.1f s" % (args.log_n, dt),
            "gpu_proof_bit_identical": bool(same),
        }
        cb = dict(line["cpu_baseline"], primitives=prim)
        if "latency" in detail:  # north_star's target is a latency ratio: the reference-CPU proof time over one GPU proof
            for k in ("api_prover_with_asserts", "api_prover", "batch_prover_b1"):
                detail["latency"]["speedup_vs_cpu_proof_" + k] = dt / (detail["latency"][k]["median_ms"] * 1e-3)
            line["cpu_baseline"]["speedup_one_proof_latency"] = detail["latency"]["speedup_vs_cpu_proof_batch_prover_b1"]
        if "ntt" in detail:
            cb["gpu_speedup_fft_2^11"] = prim["fft_2^11_ms"] / (detail["ntt"]["ms_2^11_x512"] / 512)
            cb["gpu_speedup_ec_lincomb_2^11"] = prim["ec_lincomb_2^11_s"] * 1e3 / (detail["msm"]["ms_4608"] / 4608)
            cp = prim.get("c", {})
            if "ntt_2^11_ms" in cp:  # against the compiled single-core restatement
                cb["gpu_speedup_vs_c_ntt_2^11"] = cp["ntt_2^11_ms"] / (detail["ntt"]["ms_2^11_x512"] / 512)
                cb["gpu_speedup_vs_c_ntt_2^20"] = cp["ntt_2^20_ms"] / detail["ntt"]["ms_2^20"]
                cb["gpu_speedup_vs_c_g1_lincomb_2^11"] = cp["g1_lincomb_2^11_ms"] / (detail["msm"]["ms_4608"] / 4608)
        detail["cpu_baseline"] = cb
    if rank == 0:
        emit(line, detail, args.detail)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    try:
        main()
    except (TimeoutError, ConnectionError) as exc:
        # a collective's deadline passed or a peer closed its socket: this rank is healthy, another one is not.  One line, the
        # launcher's exit code for "lost a peer", and no interpreter shutdown (destructors would wait for the device or the dead peer)
        import traceback

        traceback.print_exc()
        sys.stderr.write("bench.py[rank %s of %s]: giving up: %s: %s\n" % (os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"), type(exc).__name__, exc))
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(EXIT_PEER_LOST)
