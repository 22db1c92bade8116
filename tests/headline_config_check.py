#!/usr/bin/env python3
"""The headline configuration of bench.py as a parity check (run as a subprocess by tests/test_gpu_multiprocess.py: the number of
hardware queues is fixed when the HIP runtime starts, so pytest's own process cannot host it).

BASELINE configs[1] the way the bench line runs it: several contexts (HIP streams) on as many hardware queues, bench.py's table
budget (180 GB: the comb of 21 teeth with top tables over the 2^11 SRS bases, 12.15 additions per base), ONE workgroup per MSM (`msm_configure(0, 1)`), two
lock-step batches of 512 proofs per context, every batch of the step in flight before the first download — twice over, so that a
batch also runs behind its stream's previous one.  Checks, bit for bit:
  * every proof of the step against the SAME witnesses proved on the bucket method (the path of `ec_lincomb`, curve.py:38-111,
    no table) on a context of its own, one batch at a time;
  * the second pass of the step against the first;
  * proofs 0 and 1 against the committed fixtures (tests/golden/oracle_proofs.json: oracle proofs of the chain circuit);
  * four random proofs under the verifier's pairing check (test.py:103-133 verifies what it proves).
Prints one JSON line; exits non-zero on the first mismatch."""
import json
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tools"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8       # contexts = hardware queues
    PER = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # lock-step batches per context
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    os.environ["GPU_MAX_HW_QUEUES"] = str(NS)               # before the HIP runtime starts (bench.py does the same)
    import bench
    from plonkathon_amd import BatchProver, Context, Program, Setup
    from plonkathon_amd.batch import _pack_witnesses
    from plonkathon_amd.field import R_MOD

    bench.GROUP_ORDER = 2048
    S = NS * PER
    ctxs = [Context(0) for _ in range(NS)]
    for c in ctxs:
        c.msm_lookup(0, 0, int(bench.DEFAULT_TABLE_GB * 1e9))
        c.msm_configure(0, 1)
    setup = Setup.from_file(bench.PTAU)
    program = Program(bench.chain_program_lines(2048), 2048)
    provers = [BatchProver(setup, program, ctxs[k % NS]) for k in range(S)]
    t0 = time.perf_counter()
    blobs = []
    for k, pr in enumerate(provers):
        wits = [bench.witness_for(k * B + j) for j in range(B)]
        blobs.append(_pack_witnesses(wits, pr.variables, R_MOD))
        pr.upload_values(blobs[-1], B)
    t_wit = time.perf_counter() - t0
    passes = []
    for _ in range(2):
        t0 = time.perf_counter()
        for pr in provers:
            pr.run()                      # all S batches in flight: NS streams, PER batches queued on each
        raw = [pr.download_raw() for pr in provers]
        dt = time.perf_counter() - t0
        assert not any(any(st) for _, st in raw), "a proof of the step reported a failure status"
        passes.append(([r[0] for r in raw], dt))
    assert passes[0][0] == passes[1][0], "the second pass of the step differs from the first"
    got = passes[0][0]
    info = setup.device_bases(ctxs[0]).lookup_info()  # (the table is built by the first MSM that asks for it)
    assert info["layout"] == "comb" and info["bits"] == 21 and info["additions_per_base"] == 12 and info["top_group"] == 7 and info["sharers"] >= NS, info

    # the bucket method on the same witnesses, alone on the chip, one batch at a time
    cb = Context(0)
    cb.msm_lookup(1)
    pb = BatchProver(setup, program, cb)
    assert setup.device_bases(cb).lookup_info()["bits"] == 0
    t0 = time.perf_counter()
    for k, blob in enumerate(blobs):
        pb.upload_values(blob, B)
        pb.run()
        want, st = pb.download_raw()
        assert not any(st)
        if want != got[k]:
            bad = [j for j in range(B) if want[768 * j:768 * (j + 1)] != got[k][768 * j:768 * (j + 1)]]
            sys.exit("headline_config_check: batch %d differs from the bucket method in proofs %s" % (k, bad[:8]))
    t_bucket = time.perf_counter() - t0

    fx = {c["name"]: c for c in json.load(open(os.path.join(REPO, "tests", "golden", "oracle_proofs.json")))["cases"]}
    for i, name in ((0, "chain_2048_x0_3"), (1, "chain_2048_x0_4")):
        assert bench.proof_matches_fixture(BatchProver.decode(got[0][768 * i:768 * (i + 1)]), name), name

    rng = random.Random(20)
    vk = setup.verification_key(program.common_preprocessed_input())
    idx = sorted(rng.sample(range(S * B), 4))
    for i in idx:
        proof = BatchProver.decode(got[i // B][768 * (i % B):768 * (i % B + 1)])
        wit = bench.witness_for(i)
        assert vk.verify_proof(2048, proof, [wit[v] for v in program.get_public_assignments()]), "proof %d fails the pairing check" % i
        rec = bytearray(got[i // B][768 * (i % B):768 * (i % B + 1)])
        rec[576] ^= 1                     # one bit of the evaluation a(zeta)
        bad = BatchProver.decode(bytes(rec))
    assert not vk.verify_proof(2048, bad, [wit[v] for v in program.get_public_assignments()]), "a corrupted proof was accepted"
    print(json.dumps({"contexts": NS, "hw_queues": int(os.environ["GPU_MAX_HW_QUEUES"]), "batches": S, "batch": B, "proofs": S * B,
                      "comb_teeth": info["bits"], "comb_columns": info["additions_per_base"], "top_group": info["top_group"], "table_bytes": info["bytes"], "workgroups_per_msm": 1,
                      "identical_to_bucket_method": S * B, "fixtures": 2, "pairing_checked": idx,
                      "proofs_per_s_pass": [S * B / p[1] for p in passes], "witness_s": t_wit, "bucket_method_s": t_bucket}), flush=True)


if __name__ == "__main__":
    main()
