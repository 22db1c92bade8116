"""The parity tests proper: plonkathon_amd (Python host layer -> C-ABI -> HIP kernels on an
MI355X) against the oracle and the reference's golden vectors.  Run with `pytest -m gpu`."""
import os

import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from plonkathon_amd import Setup

    return Setup.from_file(pc.PTAU)


def test_native_library_is_loaded():
    """The HIP path is the one that runs: the in-tree .so is mapped into this process."""
    from plonkathon_amd import _lib, get_context

    name = get_context().name()
    assert "gfx950" in name, name
    with open("/proc/self/maps") as f:
        assert "libplonk_hip.so" in f.read()
    assert _lib.lib().plonk_abi_version() == 2


def test_ntt_vs_oracle():
    pc.ntt_vs_oracle([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])


def test_ntt_multipass_plans():
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        for tile, single, radix, log_ns in ((4, 2, 2, (3, 5, 8)), (6, 4, 4, (9, 12, 14)), (12, 8, 6, (13, 16)), (10, 9, 5, (15,))):
            check(ctx.L.plonk_ntt_configure(ctx.handle, tile, single, radix))
            pc.ntt_vs_oracle(log_ns, seed0=100 * tile)
    finally:
        check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))


def test_ntt_forced_variants():
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        for kind in (1, 4, 5, 6, 7, 8):  # (6 / 7: the wave kernels without / with their two-element latency forms; 8: 2^12 on 1024 threads)
            check(ctx.L.plonk_ntt_select_kernel(ctx.handle, kind))
            check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))
            pc.ntt_vs_oracle([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16], seed0=10 * kind)
            pc.ntt_roundtrip_and_linearity(20)
            check(ctx.L.plonk_ntt_configure(ctx.handle, 6, 4, 4))
            pc.ntt_vs_oracle((9, 12, 14), seed0=600 + kind)
    finally:
        check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))
        check(ctx.L.plonk_ntt_select_kernel(ctx.handle, 0))


def test_ntt_extreme_inputs():
    with pc.ntt_kind(6):  # the four- and eight-element kernels (a lone 2^9 or 2^16 would otherwise take the two-element forms)
        pc.ntt_extreme_inputs((8, 9, 10, 11, 12, 13, 16))
        pc.ntt_extreme_limbs((8, 9, 10, 11, 13))
        pc.ntt_extreme_limbs((12,), slots=8)  # 2^12 = 512 threads x 8 elements
    with pc.ntt_kind(8):  # ... and its 1024-thread, 4-element form
        pc.ntt_extreme_inputs((12,))
        pc.ntt_extreme_limbs((12,))
        pc.bls_ntt_vs_oracle((12,), seed0=277, batch=5)
    pc.ntt_extreme_inputs((14, 15, 16))  # ... and the default: latency forms


@pytest.mark.gpu
def test_ntt_latency_forms():
    """The two-element kernels alone and as both passes of 2^14 .. 2^18, BN254 and BLS12-381, forced and by default."""
    pc.ntt_latency_forms()
    with pc.ntt_kind(7):
        pc.bls_ntt_vs_oracle((7, 9), seed0=177, batch=4)
        pc.bls_ntt_vs_oracle((14, 15, 16, 17, 18), seed0=470, batch=2)
    with pc.ntt_kind(6):
        pc.ntt_two_pass_exact((14, 15, 16, 17, 18), seed0=4900)


@pytest.mark.parametrize("log_n", [18, 20, 22, 24])
def test_ntt_large_properties(log_n):
    """BASELINE microbench sizes: round trip, linearity, DC/Nyquist bins, delta -> root table."""
    pc.ntt_roundtrip_and_linearity(log_n)


def test_poly_golden():
    pc.poly_golden(max_log_n=16)


def test_round_kernels_vs_oracle():
    pc.round_kernels_vs_oracle((3, 4, 6, 9))


def test_poly_asserts():
    pc.poly_asserts()


def test_transcript():
    pc.transcript_golden()


def test_setup_commit_k1_and_vkeys(setup):
    pc.setup_k1()
    pc.vkey_goldens(setup)


def test_lincomb_golden(setup):
    pc.lincomb_golden(setup, full_size=True)


def test_msm_vs_oracle_sizes(setup):
    for n in (1, 2, 3, 63, 64, 65, 255, 1000):
        pc.msm_vs_oracle(setup, n, seed=n)


def test_msm_linearity_full_size(setup):
    pc.msm_linearity(setup, 2048)


def test_lincomb_fuzz_both_methods(setup):
    from plonkathon_amd import get_context

    ctx = get_context()
    pc.lincomb_fuzz(setup, 25)  # arbitrary bases: bucket method
    try:
        ctx.msm_lookup(2, 9)    # the same through forced lookup tables
        pc.lincomb_fuzz(setup, 25, seed=99)
    finally:
        ctx.msm_lookup(0)


def test_msm_window_configs(setup):
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        ctx.msm_lookup(1)  # the bucket method (the lookup table would otherwise serve every SRS commitment)
        for c, groups in ((4, 1), (5, 3), (6, 0), (7, 2), (8, 1), (8, 32), (9, 4), (11, 0), (12, 1), (12, 64), (13, 3)):
            check(ctx.L.plonk_msm_configure(ctx.handle, c, groups))
            pc.msm_vs_oracle(setup, 300, seed=20 + c)
            pc.msm_extreme_scalars(setup)
    finally:
        check(ctx.L.plonk_msm_configure(ctx.handle, 0, 0))
        ctx.msm_lookup(0)


def test_msm_lookup_tables():
    """Table MSM at several table sizes (forced; comb tables, then the window tables of rounds 2 - 5), then the automatic choice
    (the library's default budget, 1/16 of the device's memory: the comb of 17 teeth for the 2^11 bases on an MI355X)."""
    from plonkathon_amd import Setup, get_context

    ctx = get_context()
    try:
        for c, groups, windows in ((6, 0, False), (11, 3, False), (14, 1, False), (17, 0, False), (17, 2, False), (6, 0, True), (11, 3, True), (14, 1, True)):
            ctx.msm_lookup(2, c, 0, windows=windows)
            ctx.msm_configure(0, groups)
            s = Setup.from_file(pc.PTAU)
            pc.msm_vs_oracle(s, 300, seed=60 + c)
            pc.msm_vs_oracle(s, 2048, seed=80 + c, batch=2)
            pc.msm_extreme_scalars(s)
            pc.lincomb_golden(s, full_size=(c == 11))
            info = s.device_bases().lookup_info()
            assert info["layout"] == ("windows" if windows else "comb") and info["bits"] == c, info
            del s
    finally:
        ctx.msm_lookup(0)
        ctx.msm_configure(0, 0)
    s = Setup.from_file(pc.PTAU)
    pc.msm_vs_oracle(s, 2048, seed=99)
    pc.lincomb_golden(s, full_size=True)
    info = s.device_bases().lookup_info()
    assert info["layout"] == "comb" and info["additions_per_base"] <= 16, info  # (17 teeth, 15 additions, on a 288 GB device)


def test_msm_comb_table_shapes():
    """msm_comb_kernel's lane partition over tooth counts, base counts and workgroups per MSM (the 13 columns of the 20-tooth
    comb are covered at full size by test_full_size_lookup_table_on_an_explicit_budget and by bench.py's own verification)."""
    pc.comb_table_shapes([(2, 5, 0), (3, 40, 1), (4, 257, 1), (5, 300, 2), (7, 511, 1), (8, 256, 1), (9, 700, 1), (9, 33, 8), (12, 1000, 0),
                          (14, 2048, 1), (16, 700, 4), (19, 300, 1), (20, 120, 1), (20, 20, 0), (17, 1, 0), (17, 3, 1), (20, 2, 0), (12, 8192, 0), (13, 4097, 1)])  # (the last two: beyond 2^11 bases)


def test_msm_comb_top_tables():
    """Combs with top tables (msm_comb.h): floor(254 / h) columns plus a joint table per group of g bases for the one or two bits
    left over, dealt to the columns as virtual scalars — R = 2 with g = 1 .. 6 (h = 4 .. 18), R = 1 with g = 6 (h = 11); ragged last
    groups, fewer bases than a group, virtual scalars alone in an MSM's last workgroup, more than 2^11 bases."""
    pc.comb_table_shapes([(4, 40, 0, 1), (7, 1, 0, 1), (7, 511, 1, 1), (9, 700, 1, 1), (9, 33, 8, 1), (11, 300, 2, 1), (11, 7, 1, 1), (12, 1000, 0, 1),
                          (14, 2048, 1, 1), (14, 257, 4, 1), (18, 64, 1, 1), (12, 4097, 1, 1)])


def test_lookup_and_bucket_methods_agree():
    """512 commitments of 2^11 coefficients: byte-identical from the lookup table and from the bucket method."""
    import ctypes

    from plonkathon_amd import Setup, get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    n, M = 2048, 512
    sc = ctx.upload_ints(pc.rand_vec(4242, 4096))
    buf = ctx.alloc(n * M + 4096)
    for off in range(0, n * M + 4096, 4096):
        check(ctx.L.plonk_mem_d2d(ctx.handle, buf.at(off), sc.ptr, 32 * 4096))
    out = []
    try:
        for mode in (0, 1):
            ctx.msm_lookup(mode)
            s = Setup.from_file(pc.PTAU)
            xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
            check(ctx.L.plonk_g1_msm(ctx.handle, s.device_bases().handle, buf.ptr, n, M, n + 7, xy, fl))
            out.append((xy.raw, fl.raw))
            del s
    finally:
        ctx.msm_lookup(0)
    assert out[0] == out[1] and not any(out[0][1])


def test_batch_prover_on_the_bucket_method():
    from plonkathon_amd import Setup, get_context

    ctx = get_context()
    try:
        ctx.msm_lookup(1)
        pc.batch_prover_k6(Setup.from_file(pc.PTAU))
    finally:
        ctx.msm_lookup(0)


def test_prover_k6_golden_proof(setup):
    pc.prover_k6(setup)


def test_prover_factorization(setup):
    pc.prover_factorization(setup)


def test_deterministic_rerun(setup):
    """Run twice, compare bytes (LDS atomics make the bucket order non-deterministic; the result must not be)."""
    from plonkathon_amd import Basis

    sc = pc.rand_vec(777, 2048)
    a = setup.commit_coeffs(pc.P(sc, Basis.MONOMIAL))
    b = setup.commit_coeffs(pc.P(sc, Basis.MONOMIAL))
    assert a == b


def test_batch_prover_k6(setup):
    pc.batch_prover_k6(setup)


def test_batch_prover_public_input_counts(setup):
    pc.batch_prover_public_input_counts(setup)


def test_batch_prover_vs_oracle_small(setup):
    pc.batch_prover_vs_oracle(setup, pc.FACTORIZATION, 16, [pc.FACTORIZATION_START])
    pc.batch_prover_vs_oracle(setup, pc.chain_lines(32), 32, [{"x0": 3}, {"x0": 4}, {"x0": 12345678901234567890}])
    pc.batch_prover_vs_oracle(setup, pc.chain_lines(256), 256, [{"x0": 7}])
    pc.batch_prover_rejects_bad_witness(setup)


def test_batch_prover_group_order_2_11(setup):
    """BASELINE configs[1]: group_order = 2^11 on the powers-of-tau SRS; bit-identical proof + challenges."""
    pc.batch_prover_fixture_cases(setup, ["chain_2048_x0_3", "chain_2048_x0_4"], batch_copies=3)


def test_batch_prover_poseidon(setup):
    """BASELINE configs[2]: the mini-Poseidon circuit of test.py:216-259 at 2^10 (reference) and 2^11."""
    pc.batch_prover_fixture_cases(setup, ["poseidon_1024", "poseidon_2048"])


def test_api_prover_matches_batch_prover_2_11(setup):
    """The reference-shaped Prover (per-proof challenge as coset offset) and the lock-step prover
    (fixed offset) produce the same proof: the schedule does not change the committed polynomials."""
    from plonkathon_amd import BatchProver, Program, Prover

    program = Program(pc.chain_lines(2048), 2048)
    wit = program.fill_variable_assignments({"x0": 3})
    p1 = Prover(setup, program)
    p1.check = False
    assert pc.flat(p1.prove(dict(wit))) == pc.flat(BatchProver(setup, program).prove(dict(wit)))


@pytest.mark.parametrize("log_n", [16, 17, 18, 19, 20, 21, 22])
def test_ntt_exact_vs_c_oracle(log_n):
    """Bit-exact forward and inverse transforms at microbench sizes against the C half of the oracle."""
    from oracle import c_oracle
    from plonkathon_amd import Basis

    v = pc.rand_vec(4000 + log_n, 1 << log_n)
    assert pc.ints(pc.P(v, Basis.MONOMIAL).fft()) == c_oracle.fr_ntt(v)
    assert pc.ints(pc.P(v, Basis.LAGRANGE).ifft()) == c_oracle.fr_ntt(v, True)


def _random_canonical_bytes(seed, n):
    """n canonical elements (< 2^253 < r) as 32-byte little-endian words, from a seeded numpy generator."""
    import numpy as np

    a = np.random.default_rng(seed).integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x1F
    return a.tobytes()


@pytest.mark.parametrize("log_n,split", [(24, 0), (24, 13), (24, 12), (23, 0)])
def test_ntt_exact_vs_c_oracle_at_full_size(log_n, split):
    """BASELINE configs[3]'s largest size (and its neighbours), bit-exact in both directions against the C oracle, on bytes
    (16 M Python ints would take minutes); 2^24 on three splits (2^11 x 2^13, the default; 2^13 x 2^11; 2^12 x 2^12, whose column
    pass reads its inter-pass twiddles from the 1.3 GB table), 2^23 on its default (2^10 x 2^13, on the table as well)."""
    import ctypes

    from oracle import c_oracle
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    n = 1 << log_n
    raw = _random_canonical_bytes(9000 + log_n, n)
    try:
        if split:
            check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, split))
        buf = ctx.upload_bytes(raw)
        out = ctx.alloc(n)
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 0, 1))
        fwd = ctx.download_bytes(out)
        assert fwd == c_oracle.fr_ntt_bytes(raw), "forward"
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, buf.ptr, log_n, 1, 1))  # inverse, in place
        assert ctx.download_bytes(buf) == c_oracle.fr_ntt_bytes(raw, True), "inverse"
    finally:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, 0))


def test_ntt_inter_pass_twiddles_from_the_small_tables():
    """Two-pass transforms with no budget for the full inter-pass table (plonk_ntt_set_table_budget(0): two factors per element
    from the 1 K and N / 1 K tables): 2^18 and 2^20 exact in both directions; the default path is every other test."""
    from oracle import c_oracle
    from plonkathon_amd import Basis, get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        check(ctx.L.plonk_ntt_set_table_budget(ctx.handle, 0))
        for log_n in (18, 20):
            v = pc.rand_vec(4400 + log_n, 1 << log_n)
            assert pc.ints(pc.P(v, Basis.MONOMIAL).fft()) == c_oracle.fr_ntt(v)
            assert pc.ints(pc.P(v, Basis.LAGRANGE).ifft()) == c_oracle.fr_ntt(v, True)
        # a budget nothing new fits into: tables built by earlier tests keep serving, a size without one falls back
        check(ctx.L.plonk_ntt_set_table_budget(ctx.handle, 1))
        v = pc.rand_vec(4419, 1 << 19)
        assert pc.ints(pc.P(v, Basis.MONOMIAL).fft()) == c_oracle.fr_ntt(v)
    finally:
        check(ctx.L.plonk_ntt_set_table_budget(ctx.handle, 4 << 30))


def test_ntt_2_14_and_2_15():
    """2^14 = 2^7 x 2^7 and 2^15 = 2^7 x 2^8 on the wave kernels (two-element column pass), with the fused coset forms."""
    pc.ntt_quad_sizes()
    pc.bls_ntt_vs_oracle((14, 15), seed0=55, batch=3)


def test_bls12_381_coset_transforms():
    """coset_extend / coset_to_coeffs over the BLS12-381 scalar field at n = 2^8 .. 2^13 (batched), 2^14 -> 2^16 and 2^18 -> 2^20."""
    pc.bls_coset_vs_oracle((8, 9, 10, 11, 12, 13), batch=3)
    pc.bls_coset_vs_oracle((14, 18), seed0=270)


def test_bls12_381_ntt_every_wave_kernel():
    """The standalone BLS12-381 Fr transform, one size per wave kernel plus a two-pass size: random and extreme inputs, both
    directions, in place, batched, bad inputs refused — bit-exact against the C oracle's oracle_bls_fr_ntt."""
    with pc.ntt_kind(6):
        pc.bls_ntt_vs_oracle((8, 9, 10, 11, 12, 13), batch=5)
    pc.bls_ntt_vs_oracle((16, 17, 19), seed0=300)


@pytest.mark.parametrize("log_n", [20, 22, 24])
def test_bls12_381_ntt_exact_at_microbench_sizes(log_n):
    """BASELINE configs[3]'s sizes over the field its metric is quoted on: bit-exact forward, inverse in place, on bytes."""
    import numpy as np

    from oracle import c_oracle
    from plonkathon_amd import bls12_381 as bls

    n = 1 << log_n
    a = np.random.default_rng(7000 + log_n).integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x3F  # < 2^254 < r
    raw = a.tobytes()
    d = bls.upload(raw)
    assert bls.download(bls.ntt(d, log_n)) == c_oracle.fr_ntt_bytes(raw, False, "bls12_381"), "forward"
    bls.ntt(d, log_n, True, out=d)
    assert bls.download(d) == c_oracle.fr_ntt_bytes(raw, True, "bls12_381"), "inverse"


def test_batched_msm_vs_c_oracle(setup):
    """A batch of 12 full-size MSMs over the shared SRS (the shape rounds 1-5 launch) against the C oracle."""
    import ctypes
    from oracle import c_oracle
    from oracle.srs import Setup as OSetup
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    n, M = 2048, 12
    scal = [pc.rand_vec(9000 + m, n) for m in range(M)]
    scal[3] = [0] * n
    scal[4] = [1] + [0] * (n - 1)
    scal[5] = [pc.R_MOD - 1] * n
    buf = ctx.upload_ints([x for s in scal for x in s])
    xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
    check(ctx.L.plonk_g1_msm(ctx.handle, setup.device_bases().handle, buf.ptr, n, M, n, xy, fl))
    P = OSetup.from_file(pc.PTAU).powers_of_x
    for m in range(M):
        got = None if fl.raw[m] else (int.from_bytes(xy.raw[64 * m : 64 * m + 32], "little"),
                                      int.from_bytes(xy.raw[64 * m + 32 : 64 * m + 64], "little"))
        assert got == c_oracle.g1_lincomb(P, scal[m]), m


def test_edge_and_error_paths(setup):
    pc.edge_and_error_paths(setup)


def test_two_streams_share_the_gpu(setup):
    """Two lock-step provers on two contexts (HIP streams) of one GPU run concurrently and stay bit-exact."""
    from plonkathon_amd import BatchProver, Context, Program

    program = Program(pc.chain_lines(256), 256)
    wits = [program.fill_variable_assignments({"x0": 3 + i}) for i in range(6)]
    ctx2 = Context(0)
    p1, p2 = BatchProver(setup, program), BatchProver(setup, program, ctx2)
    p1.upload(wits[:3])
    p2.upload(wits[3:])
    for _ in range(2):
        p1.run()
        p2.run()
    got = [pc.flat(p) for p in p1.download() + p2.download()]
    ref = BatchProver(setup, program)
    assert got == [pc.flat(p) for p in ref.prove_batch(wits)]


def test_gpu_proofs_verify_under_the_pairing_check(setup):
    """group_order 2^11 (chain) and the Poseidon circuit at 2^11: proofs verify (TESTING_verifier's equations)."""
    from oracle.poseidon import poseidon_program_lines

    pc.proofs_verify(setup, pc.chain_lines(2048), 2048, {"x0": 5}, ["x0"])
    pc.proofs_verify(setup, pc.chain_lines(2048), 2048, {"x0": 0xDEADBEEF12345}, ["x0"])  # a second witness per circuit
    pc.proofs_verify(setup, poseidon_program_lines(), 2048, {"L0": 1, "M0": 2}, ["L0", "M0", "M64"])
    pc.proofs_verify(setup, poseidon_program_lines(), 2048, {"L0": 77, "M0": 123456789}, ["L0", "M0", "M64"])


@pytest.mark.gpu
def test_batch_prover_tiny_group_orders_and_resident_batch(setup):
    pc.batch_prover_tiny_group_orders(setup)
    pc.batch_prover_resident_batch_is_checked(setup)


@pytest.mark.gpu
def test_msm_deferred_overflow_is_recomputed():
    pc.msm_deferred_overflow()


@pytest.mark.gpu
def test_lagrange_srs_by_group_ntt_equals_the_msm_route():
    pc.lagrange_srs_by_ntt((0, 3, 8, 11))
    pc.lagrange_srs_beyond_2e12(13)
    pc.lagrange_srs_beyond_2e12(16)  # the largest size the group transform is covered at (kzg.LAGRANGE_SRS_MAX_LOG caps the view at 2^20)


@pytest.mark.gpu
def test_lookup_table_is_shared_across_contexts():
    pc.lookup_table_is_shared_across_contexts()
    pc.lookup_table_colliding_key()


@pytest.mark.gpu
def test_configs4_batch_of_512_distinct_proofs_sharded_and_gathered(setup):
    """BASELINE configs[4]: 512 independent group_order = 2^11 proofs.  One lock-step batch of 512 distinct witnesses:
    proofs 0 and 1 equal the fixtures, every status is 0, 16 random ones pass the pairing check (a corrupted one does not),
    2 random ones equal the oracle prover's, and the batch is byte-identical to the same 512 proofs
    proved as 8 shards of 64 (the 8-GPU sharding, rank r owning indices r, r+8, ...) and reassembled by
    distributed.gather_proofs."""
    import json

    import bench
    from plonkathon_amd import BatchProver, Program
    from plonkathon_amd import distributed as D

    n, total, world = 2048, 512, 8
    bench.GROUP_ORDER = n
    program = Program(pc.chain_lines(n), n)
    wits = [bench.witness_for(i) for i in range(total)]
    bp = BatchProver(setup, program)
    bp.upload(wits)
    bp.run()
    blob, status = bp.download_raw()
    assert len(blob) == 768 * total and not any(status)
    fx = {c["name"]: c for c in json.load(open(os.path.join(pc.GOLDEN, "oracle_proofs.json")))["cases"]}
    for b, name in ((0, "chain_2048_x0_3"), (1, "chain_2048_x0_4")):
        got = pc.flat(BatchProver.decode(blob[768 * b : 768 * (b + 1)]))
        for k, v in fx[name]["proof"].items():
            assert got[k] == (pc.pt(v) if isinstance(v, list) else int(v)), (name, k)
    assert len({blob[768 * i : 768 * (i + 1)] for i in range(total)}) == total  # all distinct

    # Independent acceptance of the bulk (the reference verifies what it proves: test.py:103-133).  16 random indices under the
    # pairing check of VerificationKey.verify_proof, 2 random indices byte for byte against the oracle's prover (a CPU proof
    # each), and a proof corrupted at a random index must be rejected.
    import random

    from oracle.circuit import Program as OProgram
    from oracle.plonk_prover import Prover as OProver
    from oracle.srs import Setup as OSetup

    rng = random.Random(20260928)
    vk = setup.verification_key(program.common_preprocessed_input())
    picked = rng.sample(range(2, total), 16)
    for i in picked:
        assert vk.verify_proof(n, BatchProver.decode(blob[768 * i : 768 * (i + 1)]), [wits[i]["x0"]]), i
    i = picked[0]
    bad = bytearray(blob[768 * i : 768 * (i + 1)])
    bad[rng.randrange(768 - 6 * 32, 768)] ^= 0x01          # one bit of one of the six evaluations
    assert not vk.verify_proof(n, BatchProver.decode(bytes(bad)), [wits[i]["x0"]])
    bad = bytearray(blob[768 * i : 768 * (i + 1)])
    bad[64 * rng.randrange(9) + 1] ^= 0x80                  # a commitment pushed off its value (and the curve)
    try:
        accepted = vk.verify_proof(n, BatchProver.decode(bytes(bad)), [wits[i]["x0"]])
    except Exception:  # (the library may refuse a point that is not on the curve outright)
        accepted = False
    assert not accepted
    assert not vk.verify_proof(n, BatchProver.decode(blob[768 * i : 768 * (i + 1)]), [wits[i]["x0"] + 1])  # wrong public input
    oprover = OProver(OSetup.from_file(os.path.join(pc.GOLDEN, "srs_2048.ptau")), OProgram(pc.chain_lines(n), n))
    for i in rng.sample(range(2, total), 2):
        want = oprover.prove(dict(wits[i])).flatten()
        got = pc.flat(BatchProver.decode(blob[768 * i : 768 * (i + 1)]))
        for k, v in want.items():
            assert got[k] == v, (i, k)

    shards = []
    sp = BatchProver(setup, program)
    for r in range(world):
        idx = D.shard_indices(total, r, world)
        sp.upload([wits[i] for i in idx])
        sp.run()
        sb, st = sp.download_raw()
        assert not any(st)
        shards.append(sb)

    class ReplayComm:  # the all-gather of 8 ranks, replayed in one process
        def __init__(self, rank):
            self.rank, self.world = rank, world

        def all_gather(self, payload):
            assert payload == shards[self.rank]
            return shards

    for r in (0, 5):
        assert b"".join(D.gather_proofs(shards[r], total, ReplayComm(r))) == blob


@pytest.mark.gpu
def test_rccl_gather_through_the_c_abi():
    """plonk_comm_* / plonk_gather_results on real RCCL (one rank: this box has one GPU; the 8-GPU run is the driver's)."""
    from plonkathon_amd import get_context
    from plonkathon_amd import distributed as D

    comm = D.RcclComm(get_context(), 0, 1)
    payload = bytes(range(256)) * 3
    assert comm.all_gather(payload) == [payload]
    assert comm.max(1.25) == 1.25
    comm.barrier()
    assert D.gather_proofs(payload, 1, comm) == [payload]
    comm.close()


@pytest.mark.gpu
def test_lagrange_srs_paths(setup):
    pc.lagrange_srs_paths(setup)
    # full size: the 2^11 chain circuit proved with Lagrange-basis commitments equals the fixture
    from plonkathon_amd import BatchProver, Program
    import bench, json

    bench.GROUP_ORDER = 2048
    bp = BatchProver(setup, Program(pc.chain_lines(2048), 2048), lagrange_commits=True)
    proofs = bp.prove_batch([bench.witness_for(0), bench.witness_for(1)])
    fx = {c["name"]: c for c in json.load(open(os.path.join(pc.GOLDEN, "oracle_proofs.json")))["cases"]}
    for b, name in ((0, "chain_2048_x0_3"), (1, "chain_2048_x0_4")):
        got = pc.flat(proofs[b])
        for k, v in fx[name]["proof"].items():
            assert got[k] == (pc.pt(v) if isinstance(v, list) else int(v)), (name, k)


@pytest.mark.gpu
def test_async_upload_and_device_resident_gather(setup):
    pc.async_upload_and_device_gather(setup, rccl=True)


def test_g1_and_proof_encoding(setup):
    pc.g1_encoding_cases(setup)


def test_product_verifier(setup):
    pc.verifier_cases(setup, full_size=True)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [18, 22])
def test_distributed_ntt_single_rank_rccl(log_n):
    """plonk_fr_ntt_distributed through a real (one-rank) RCCL communicator: column pass, all-to-all, row pass; the output
    layout [R2][R1] holds frequency k1 + R1 k2 at [k2][k1] — for one rank that is the natural order.  Forward and inverse."""
    from oracle import c_oracle
    from plonkathon_amd import get_context
    from plonkathon_amd import distributed as D

    ctx = get_context()
    comm = D.RcclComm(ctx, 0, 1)
    v = pc.rand_vec(900 + log_n, 1 << log_n)
    d_in, d_out = ctx.upload_ints(v), ctx.alloc(len(v))
    D.ntt_distributed(comm, ctx, d_in, d_out, log_n)
    assert ctx.download_ints(d_out) == c_oracle.fr_ntt(v)
    D.ntt_distributed(comm, ctx, d_in, d_out, log_n, inverse=True)
    assert ctx.download_ints(d_out) == c_oracle.fr_ntt(v, True)
    comm.close()


@pytest.mark.gpu
def test_full_size_lookup_table_on_an_explicit_budget():
    """The comb of 20 teeth (68.7 GB, 13 additions per base: what bench.py opts into with its 100 GB budget) — the default budget
    never builds it.  The 2^11 chain proofs against the fixtures, both methods byte-identical on 8 more proofs, and the table
    really is the one attached (layout, bits, bytes, shared by a second context)."""
    import json

    import bench
    from plonkathon_amd import BatchProver, Context, Program, Setup

    a, b = Context(0), Context(0)
    a.msm_lookup(0, 0, int(100e9))
    b.msm_lookup(0, 0, int(100e9))
    sa = Setup.from_file(pc.PTAU)
    bench.GROUP_ORDER = 2048
    program = Program(pc.chain_lines(2048), 2048)
    proofs = BatchProver(sa, program, a).prove_batch([bench.witness_for(0), bench.witness_for(1)])
    info = sa.device_bases(a).lookup_info()
    assert info["layout"] == "comb" and info["bits"] == 20 and info["additions_per_base"] == 13 and info["bytes"] == 2048 * (1 << 19) * 64, info
    fx = {c["name"]: c for c in json.load(open(os.path.join(pc.GOLDEN, "oracle_proofs.json")))["cases"]}
    for i, name in ((0, "chain_2048_x0_3"), (1, "chain_2048_x0_4")):
        got = pc.flat(proofs[i])
        for k, v in fx[name]["proof"].items():
            assert got[k] == (pc.pt(v) if isinstance(v, list) else int(v)), (name, k)
    # a second context on the same device attaches to the same table; the bucket method agrees byte for byte
    pb = BatchProver(sa, program, b)
    wits = [bench.witness_for(10 + i) for i in range(8)]
    pb.upload(wits)
    pb.run()
    blob_lookup = pb.download_raw()[0]
    assert sa.device_bases(b).lookup_info()["bits"] == 20 and sa.device_bases(b).lookup_info()["sharers"] == 2
    c = Context(0)
    c.msm_lookup(1)
    pcx = BatchProver(sa, program, c)
    pcx.upload(wits)
    pcx.run()
    assert pcx.download_raw()[0] == blob_lookup
    del pb, pcx, proofs


@pytest.mark.gpu
def test_full_size_comb_with_top_tables():
    """What a 180 GB budget buys (bench.py's opt-in): the comb of 21 teeth with top tables over the 2^11 SRS bases — 12 columns and a
    joint table per 7 bases for the two bits left over: 12 x 2 073 additions per MSM of 2^11 (12.15 per base; 13 on the 20-tooth
    comb), 2 348 blocks of 2^20 entries = 157.6 GB.  The 2^11 chain proofs against the fixtures; 8 more proofs byte-identical on
    the table, on the plain 17-tooth comb and on the bucket method; commitments of fewer coefficients than the SRS has bases (the
    virtual scalars then sit further from their blocks) and of the last top bits (scalars from 2^253 up) against the oracle."""
    import json

    import bench
    from oracle import c_oracle
    from plonkathon_amd import BatchProver, Context, Program, Setup
    from plonkathon_amd.polynomial import Basis

    a = Context(0)
    a.msm_lookup(0, 0, int(180e9))
    sa = Setup.from_file(pc.PTAU)
    bench.GROUP_ORDER = 2048
    program = Program(pc.chain_lines(2048), 2048)
    proofs = BatchProver(sa, program, a).prove_batch([bench.witness_for(0), bench.witness_for(1)])
    info = sa.device_bases(a).lookup_info()
    assert info["layout"] == "comb" and info["bits"] == 21 and info["additions_per_base"] == 12 and info["top_bits"] == 2 and info["top_group"] == 7, info
    assert info["bytes"] == (2048 + 25 * 12) * (1 << 20) * 64 and sa.device_bases(a).table_additions(info, 2048) == 12 * 2073, info
    fx = {c["name"]: c for c in json.load(open(os.path.join(pc.GOLDEN, "oracle_proofs.json")))["cases"]}
    for i, name in ((0, "chain_2048_x0_3"), (1, "chain_2048_x0_4")):
        got = pc.flat(proofs[i])
        for k, v in fx[name]["proof"].items():
            assert got[k] == (pc.pt(v) if isinstance(v, list) else int(v)), (name, k)
    wits = [bench.witness_for(10 + i) for i in range(8)]
    pa_ = BatchProver(sa, program, a)
    pa_.upload(wits)
    pa_.run()
    blob = pa_.download_raw()[0]
    for conf in ((1,), (0, 17, int(20e9))):  # bucket method; the plain comb of 17 teeth
        c = Context(0)
        c.msm_lookup(*conf)
        pcx = BatchProver(sa, program, c)
        pcx.upload(wits)
        pcx.run()
        assert pcx.download_raw()[0] == blob, conf
        del pcx
    # short MSMs on the same table, top bits set
    import random

    from plonkathon_amd import backend

    rng = random.Random(606)
    prev = backend.get_context()
    backend.set_context(a)
    try:
        pts = [pc.affine(p) for p in sa.powers_of_x[:300]]
        for n in (1, 6, 7, 8, 85, 300):
            sc = [rng.choice([pc.R_MOD - 1 - rng.randrange(1 << 200), (1 << 253) + rng.randrange(1 << 250), rng.randrange(pc.R_MOD)]) for _ in range(n)]
            got = sa.commit_coeffs(pc.P(sc, Basis.MONOMIAL))
            assert pc.affine(got) == c_oracle.g1_lincomb(pts[:n], sc), n
    finally:
        backend.set_context(prev)
    del pa_, proofs


@pytest.mark.gpu
@pytest.mark.parametrize("log_n,split", [(18, 9), (18, 10), (20, 11), (20, 12), (22, 13), (22, 9), (19, 11), (21, 8)])
def test_wave_kernel_two_pass_splits_and_batches(log_n, split):
    """Two-pass wave transforms on splits the dispatcher does not pick by itself (every pairing of the 4- and
    8-element-per-thread kernels, and of small with large factors): a batch of three transforms in one call, exact
    against the C oracle, and back in place."""
    import ctypes

    from oracle import c_oracle
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, split))
        n = 1 << log_n
        raws = [_random_canonical_bytes(70 + 10 * log_n + i, n) for i in range(3)]
        buf = ctx.upload_bytes(b"".join(raws))
        out = ctx.alloc(3 * n)
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 0, 3))
        got = ctx.download_bytes(out)
        for i, raw in enumerate(raws):
            assert got[32 * n * i : 32 * n * (i + 1)] == c_oracle.fr_ntt_bytes(raw), i
        check(ctx.L.plonk_fr_ntt(ctx.handle, out.ptr, out.ptr, log_n, 1, 3))  # in place, inverse
        assert ctx.download_bytes(out) == b"".join(raws)
    finally:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, 0))


def test_bls12_381_golden_vectors():
    """The committed known-answer vectors of the BLS12-381 scalar-field transforms (definition in Python integers), 2^8 .. 2^16."""
    pc.bls_golden(max_log_n=16)
