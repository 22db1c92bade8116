"""The emulated kernels under other fibre schedules (tests/emu/hip_emu.cpp: PLONK_EMU_SCHED=reverse / random:<seed>).  Between
synchronisation points any thread order is a legal GPU schedule, so results must not depend on it: a failure here means a
missing barrier.  Runs a slice of the parity cases in a child process per schedule (the emulator reads the variable once)."""
import os
import subprocess
import sys

import pytest

from conftest import EMU_LIB, REPO

CASES = """
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import parity_cases as pc
from plonkathon_amd.kzg import Setup
pc.ntt_vs_oracle((5, 9, 11, 12), seed0=41)          # the LDS kernel (single pass), wave (limb form) and the LDS kernel's multi-pass plans
pc.ntt_extreme_inputs((9,))
setup = Setup.from_file(pc.PTAU)
pc.msm_vs_oracle(setup, 64, seed=7, batch=3)         # lookup-table MSM (tiny table budget) incl. its LDS tree
pc.batch_prover_k6(setup)                            # the whole lock-step prover: scans, transcript, divisions
print("ok")
"""


@pytest.mark.parametrize("sched", ["reverse", "random:3"])
def test_results_do_not_depend_on_the_thread_schedule(emu_cdll, sched):
    env = dict(os.environ, PLONK_HIP_LIB=EMU_LIB, PLONK_MSM_TABLE_GB="0.0001", PLONK_EMU_SCHED=sched)
    r = subprocess.run([sys.executable, "-c", CASES % (REPO, os.path.join(REPO, "tests"))], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-500:], r.stderr[-2000:])
