"""N>1 path on CPU: two gloo ranks shard a batch of proofs by index, each proves its shard through the
product's BatchProver (over the emulated kernels, injected from the test side), the 768-byte results
are all_gathered, and rank 0 checks every proof against the oracle."""
import ctypes
import os
import sys

import pytest
import torch.multiprocessing as mp

from conftest import EMU_LIB, REPO

LINES = ["e public", "c <== a * b", "e <== c * d"]
WITS = [{"a": 3 + i, "b": 4, "c": (3 + i) * 4, "d": 5, "e": (3 + i) * 20} for i in range(5)]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from plonkathon_amd import _lib

    _lib.bind(ctypes.CDLL(EMU_LIB))  # test-side injection of the emulated kernels
    from plonkathon_amd import BatchProver, Program, Setup
    from plonkathon_amd import distributed as D

    dist = D.init_from_env("gloo")
    setup = Setup.from_file(os.path.join(REPO, "tests", "golden", "srs_2048.ptau"))
    prover = BatchProver(setup, Program(LINES, 8))
    mine = D.shard_indices(len(WITS), rank, world)
    prover.upload([dict(WITS[i]) for i in mine])
    prover.run()
    blob, status = prover.download_raw()
    assert not any(status)
    allp = D.gather_proofs(blob, len(WITS), dist)
    slowest = D.max_over_ranks(float(rank), dist)
    if rank == 0:
        q.put(([p.hex() for p in allp], slowest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(emu_cdll):
    from plonkathon_amd import BatchProver
    from oracle.circuit import Program as OProgram
    from oracle.plonk_prover import Prover as OProver
    from oracle.srs import Setup as OSetup

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    hexes, slowest = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slowest == 1.0 and len(hexes) == len(WITS)
    osetup = OSetup.from_file(os.path.join(REPO, "tests", "golden", "srs_2048.ptau"))
    for w, hx in zip(WITS, hexes):
        got = BatchProver.decode(bytes.fromhex(hx)).flatten()
        want = OProver(osetup, OProgram(LINES, 8)).prove(dict(w)).flatten()
        for k, v in want.items():
            g = got[k]
            assert ((g[0].n, g[1].n) if isinstance(g, tuple) else g.n) == v, k


def test_shard_indices_cover_everything():
    from plonkathon_amd.distributed import shard_indices

    for total in (0, 1, 5, 512):
        for world in (1, 2, 4, 8):
            seen = sorted(i for r in range(world) for i in shard_indices(total, r, world))
            assert seen == list(range(total))
