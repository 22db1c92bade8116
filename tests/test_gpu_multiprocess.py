"""The 8-rank launch contract on real kernels: `bench.py --gpus 8` as eight real processes that share the
one GPU of the test box (socket transport instead of RCCL, which needs a device per rank), the library's default 4 GiB
table budget per process — launch, rendezvous, proof-index sharding, per-rank proving and the final gather are the ones
the driver's 8-GPU run uses; only the transport of the 768-byte records differs.  The 512 gathered proofs (BASELINE
configs[4]: 8 x 64) must be byte-identical to the same 512 witnesses proved as one lock-step batch in this process."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_eight_processes_share_the_gpu_and_gather_512_proofs(tmp_path):
    dump = tmp_path / "proofs.bin"
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dist-backend", "sockets", "--steps", "1", "--warmup", "0",
           "--batch", "64", "--batches-per-step", "1", "--streams", "1", "--lookup-budget-gb", "4", "--no-cpu-baseline",
           "--no-microbench", "--no-fallbacks", "--no-end-to-end", "--no-configs", "--no-latency", "--dump-proofs", str(dump)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["ranks_in_communicator"] == 8
    assert line["config"]["results_gathered_per_step"] == 512 and line["config"]["gather_in_timed_region"]
    assert line["config"]["msm_table_bits"] == 11  # the 4 GiB budget: 3.2 GB table per process
    blob = dump.read_bytes()
    assert len(blob) == 512 * 768

    sys.path.insert(0, REPO)
    import bench
    from plonkathon_amd import BatchProver, Program, Setup

    setup = Setup.from_file(bench.PTAU)
    program = Program(bench.chain_program_lines(bench.GROUP_ORDER), bench.GROUP_ORDER)
    bp = BatchProver(setup, program)
    bp.upload([bench.witness_for(i) for i in range(512)])
    bp.run()
    raw, status = bp.download_raw()
    assert not any(status)
    assert raw == blob
