#!/usr/bin/env python3
"""Exercises the RCCL leg of plonkathon_amd.distributed in one process (world_size 1) next to the HIP library:
the 8-GPU run is the driver's, this only checks that torch's RCCL and libplonk_hip.so coexist on one device."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import torch
import torch.distributed as dist

from plonkathon_amd import BatchProver, Context, Program, Setup, set_context
from plonkathon_amd import distributed as D

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = Context(0)
set_context(ctx)
setup = Setup.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "srs_2048.ptau"))
program = Program(["e public", "c <== a * b", "e <== c * d"], 8)
pr = BatchProver(setup, program, ctx)
pr.upload([program.fill_variable_assignments({"a": 3, "b": 4, "d": 5})] * 3)
pr.run()
raw, status = pr.download_raw()
assert not any(status)
torch.cuda.synchronize()
dist.barrier()
got = D.gather_proofs(raw, 3, dist)
assert b"".join(got) == raw
print("max_over_ranks", D.max_over_ranks(1.25, dist), "gathered", len(got), "proofs over", dist.get_backend())
dist.destroy_process_group()
