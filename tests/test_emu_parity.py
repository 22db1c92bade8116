"""CPU suite: the product's Python layer + C-ABI over the EMULATED build of the kernel sources
(tests/emu/).  Checks kernel index/barrier logic and the host layer without a GPU; small sizes only.
The parity tests proper are tests/test_gpu_parity.py (-m gpu)."""
import pytest

import parity_cases as pc


def test_ntt_vs_oracle_small(emu):
    pc.ntt_vs_oracle([0, 1, 2, 3, 5, 8, 9, 10, 11, 12, 13])


def test_ntt_multipass_paths(emu):
    """Forces 2-, 3- and 4-pass plans at small sizes (digit-reversing store, inter-pass twiddles)."""
    import ctypes
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        for tile, single, radix, log_ns in ((4, 2, 2, (3, 4, 5, 6, 7, 8)), (5, 3, 3, (7, 9)), (6, 4, 4, (9, 11, 12))):
            check(ctx.L.plonk_ntt_configure(ctx.handle, tile, single, radix))
            pc.ntt_vs_oracle(log_ns, seed0=100 * tile)
    finally:
        check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))


def test_ntt_forced_variants(emu):
    """The LDS kernel and the wave kernels (without / with their latency forms) forced at every size class."""
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        for kind in (1, 6, 7, 8):  # (6 / 7: the wave kernels without / with their two-element latency forms; 8: 2^12 on 1024 threads)
            check(ctx.L.plonk_ntt_select_kernel(ctx.handle, kind))
            check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))
            pc.ntt_vs_oracle([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13] if kind != 8 else [12], seed0=10 * kind)
            if kind == 8:
                continue
            check(ctx.L.plonk_ntt_configure(ctx.handle, 6, 4, 4))
            pc.ntt_vs_oracle((9, 11, 12), seed0=600 + kind)
    finally:
        check(ctx.L.plonk_ntt_configure(ctx.handle, 0, 0, 0))
        check(ctx.L.plonk_ntt_select_kernel(ctx.handle, 0))


def test_ntt_properties(emu):
    pc.ntt_roundtrip_and_linearity(12)


def test_ntt_extreme_inputs(emu):
    with pc.ntt_kind(6):  # the four- and eight-element kernels (a lone 2^9 would otherwise take its two-element form)
        pc.ntt_extreme_inputs((8, 9, 10, 11, 12, 13))
        pc.ntt_extreme_limbs((8, 9, 10, 11))
        pc.ntt_extreme_limbs((12,), slots=8)  # 2^12 = 512 threads x 8 elements
    with pc.ntt_kind(8):  # ... and its 1024-thread, 4-element form (the column-pass kernel of 2^24 = 2^12 x 2^12)
        pc.ntt_extreme_limbs((12,))


def test_ntt_latency_forms(emu):
    pc.ntt_latency_forms(two_pass=(14, 15, 16), batched=())  # (2^17, 2^18 and the batched calls: the GPU suite)


def test_bls12_381_ntt(emu):
    """One size per wave kernel (E = 2, 4 and 8; 0, 1, 2 LDS stages) with the emulator's range checks on the 255-bit modulus."""
    with pc.ntt_kind(6):
        pc.bls_ntt_vs_oracle((8, 9, 10, 11), batch=2)
        pc.bls_ntt_vs_oracle((12, 13), seed0=77)
    with pc.ntt_kind(7):
        pc.bls_ntt_vs_oracle((7, 9), seed0=177, batch=2)



def test_poly_golden(emu):
    pc.poly_golden(max_log_n=11)


def test_round_kernels_vs_oracle(emu):
    pc.round_kernels_vs_oracle((3, 5))


def test_poly_asserts(emu):
    pc.poly_asserts()


def test_transcript(emu):
    pc.transcript_golden()


@pytest.fixture(scope="module")
def setup_obj(emu_cdll):
    return {}


def test_setup_commit_vkeys_lincomb(emu):
    setup = pc.setup_k1()
    pc.vkey_goldens(setup)
    pc.lincomb_golden(setup, full_size=False)
    pc.msm_vs_oracle(setup, 200, seed=11)
    pc.msm_linearity(setup, 96)
    pc.lincomb_fuzz(setup, 12)


def test_batched_operators(emu):
    from plonkathon_amd import Setup

    pc.batched_operators(Setup.from_file(pc.PTAU), sizes=(8, 64))


def test_msm_window_configs(emu):
    from plonkathon_amd import Setup, get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    try:
        for c, groups in ((4, 1), (5, 3), (9, 2), (7, 0), (12, 1), (13, 4)):
            check(ctx.L.plonk_msm_configure(ctx.handle, c, groups))
            pc.msm_vs_oracle(Setup.from_file(pc.PTAU), 64, seed=20 + c)
            pc.msm_extreme_scalars(Setup.from_file(pc.PTAU))
    finally:
        check(ctx.L.plonk_msm_configure(ctx.handle, 0, 0))


def test_msm_comb_table_shapes(emu):
    """msm_comb_kernel's lane partition over tooth counts (a = 127 .. 29 columns), base counts and workgroups per MSM."""
    pc.comb_table_shapes([(2, 5, 0), (3, 40, 1), (4, 1, 0), (4, 257, 1), (5, 300, 2), (6, 64, 0), (7, 19, 1), (7, 511, 1), (8, 256, 1),
                          (9, 700, 1), (9, 33, 8), (13, 2, 0), (16, 1, 0), (17, 3, 1)])  # (the last three: one live piece in a column's list of 12 .. 17)


def test_msm_comb_top_tables(emu):
    """Combs with top tables (msm_comb.h: floor(254 / h) columns, the one or two bits left over through a joint table per group of
    g bases, dealt to the columns as virtual scalars): R = 2 with g = 2 (h = 7, 9) and g = 3 (h = 12), R = 1 with g = 6 (h = 11);
    fewer bases than a group, ragged last groups, several workgroups per MSM (virtual scalars in the last one), and — inside
    comb_table_shapes — scalars 0 / 1 / r - 1 / 2^253 (top bits set), an identity base, all-zero and cancelling MSMs."""
    pc.comb_table_shapes([(7, 1, 0, 1), (7, 19, 1, 1), (7, 300, 2, 1), (9, 33, 4, 1), (11, 64, 0, 1), (11, 7, 1, 1), (12, 50, 1, 1)])
    # the shape the library reports for an SRS table with top tables, and a commitment shorter than the SRS on it (the virtual scalars'
    # blocks then lie N - n blocks further from their scalars: top_delta)
    import random

    from plonkathon_amd import Basis, Setup, get_context

    ctx = get_context()
    setup = Setup.from_file(pc.PTAU)
    rng = random.Random(9)
    for h, cols, top_bits, group in ((7, 36, 2, 2), (9, 28, 2, 2)):
        ctx.msm_lookup(2, h, 0, top=True)
        n = 100
        vals = [rng.randrange(pc.R_MOD) for _ in range(n - 2)] + [pc.R_MOD - 1, 1 << 253]
        got = setup.commit_coeffs(pc.P(vals, Basis.MONOMIAL))
        assert pc.affine(got) == pc.og1.ec_lincomb(list(zip([pc.affine(p) for p in setup.powers_of_x[:n]], vals))), h
        info = setup.device_bases().lookup_info()
        assert (info["layout"], info["bits"], info["additions_per_base"], info["top_bits"], info["top_group"]) == ("comb", h, cols, top_bits, group), info
        groups = -(-2048 // group)
        assert info["bytes"] == (2048 + -(-groups // cols) * cols) * (1 << (h - 1)) * 64, info
        assert setup.device_bases().table_additions(info, n) == cols * (n + -(-(-(-n // group)) // cols)), info
    ctx.msm_lookup(0)
    # the request is refused where a comb takes no top tables (254 mod teeth not 1 or 2), without an explicit tooth count, or on window tables
    for args, kw in (((2, 20, 0), {"top": True}), ((2, 8, 0), {"top": True}), ((0, 0, 0), {"top": True}), ((2, 12, 0), {"top": True, "windows": True})):
        with pytest.raises(AssertionError):
            ctx.msm_lookup(*args, **kw)
    ctx.msm_lookup(0)


@pytest.mark.parametrize("windows", [False, True])
def test_msm_lookup_tables(emu, windows):
    """The table MSM — comb tables (csrc/msm_comb.h), and the window tables of rounds 2 - 5 (every multiple of every window
    base) — against the oracle and the goldens, at sizes small enough for the emulator; includes the identity / duplicate /
    cancelling base cases."""
    from plonkathon_amd import Setup, get_context

    ctx = get_context()
    real = ctx.msm_lookup
    ctx.msm_lookup = lambda mode=0, bits=0, budget_bytes=0: real(mode, bits, budget_bytes, windows=windows)
    try:
        big = 4 if windows else 5  # (a window table of the 2^11 bases has 255 / c times the entries of the comb: one size less there)
        for c, groups in ((3, 0), (big, 2)):
            ctx.msm_lookup(2, c)
            ctx.msm_configure(0, groups)
            setup = Setup.from_file(pc.PTAU)
            pc.msm_vs_oracle(setup, 64, seed=40 + c)
            pc.msm_extreme_scalars(setup)
            if c == big:
                pc.lincomb_golden(setup, full_size=False)
                pc.lincomb_fuzz(setup, 8, seed=77)
        # 256 lanes per MSM and at least as many scalars: the lookup kernel's batch order (lane t takes scalars t, t + 256, ..),
        # with a ragged tail (300) and an exact multiple (512), two MSMs per call
        ctx.msm_lookup(2, big)
        ctx.msm_configure(0, 1)
        setup = Setup.from_file(pc.PTAU)
        pc.msm_vs_oracle(setup, 300, seed=91)
        if not windows:  # (the window tables are the comparison layout since round 5: the GPU suite runs their full set)
            pc.msm_vs_oracle(setup, 512, seed=92)
        ctx.msm_configure(0, 0)
        ctx.msm_lookup(2, 4)
        pc.prover_k6(Setup.from_file(pc.PTAU))
        if not windows:
            pc.batch_prover_k6(Setup.from_file(pc.PTAU))
    finally:
        del ctx.msm_lookup
        ctx.msm_lookup(0)
        ctx.msm_configure(0, 0)


def test_prover_k6_golden_proof(emu):
    from plonkathon_amd import Setup

    pc.prover_k6(Setup.from_file(pc.PTAU))


def test_prover_factorization(emu):
    from plonkathon_amd import Setup

    pc.prover_factorization(Setup.from_file(pc.PTAU))


def test_batch_prover_k6(emu):
    from plonkathon_amd import Setup

    pc.batch_prover_k6(Setup.from_file(pc.PTAU))


def test_batch_prover_public_input_counts(emu):
    from plonkathon_amd import Setup

    pc.batch_prover_public_input_counts(Setup.from_file(pc.PTAU))


def test_batch_prover_vs_oracle(emu):
    from plonkathon_amd import Setup

    setup = Setup.from_file(pc.PTAU)
    pc.batch_prover_vs_oracle(setup, pc.FACTORIZATION, 16, [pc.FACTORIZATION_START])
    pc.batch_prover_vs_oracle(setup, pc.chain_lines(32), 32, [{"x0": 3}, {"x0": 4}, {"x0": 12345678901234567890}])
    pc.batch_prover_rejects_bad_witness(setup)


def test_batch_prover_on_the_wave_kernels(emu):
    """group_order 2^8: the smallest size whose transforms run on the wave kernels — the three coset evaluations of every
    wire polynomial as ONE fanned launch, the quotient's three inverse transforms likewise (smaller orders loop over calls)."""
    from plonkathon_amd import Setup

    pc.batch_prover_vs_oracle(Setup.from_file(pc.PTAU), pc.chain_lines(256), 256, [{"x0": 3}, {"x0": 77}])


def test_edge_and_error_paths(emu):
    from plonkathon_amd import Setup

    pc.edge_and_error_paths(Setup.from_file(pc.PTAU))


def test_proofs_verify_under_the_pairing_check(emu):
    from plonkathon_amd import Setup

    setup = Setup.from_file(pc.PTAU)
    pc.proofs_verify(setup, ["e public", "c <== a * b", "e <== c * d"], 8, {"a": 3, "b": 4, "d": 5}, ["e"])
    pc.proofs_verify(setup, pc.FACTORIZATION, 16, pc.FACTORIZATION_START, ["n"])


def test_batch_prover_tiny_group_orders_and_resident_batch(emu):
    from plonkathon_amd import Setup

    setup = Setup.from_file(pc.PTAU)
    pc.batch_prover_tiny_group_orders(setup)
    pc.batch_prover_resident_batch_is_checked(setup)


def test_msm_deferred_overflow_is_recomputed(emu):
    pc.msm_deferred_overflow()


def test_lagrange_srs_by_group_ntt(emu):
    pc.lagrange_srs_by_ntt((0, 1, 2, 5))
    import os

    os.environ["PLONK_LAGRANGE_SRS"] = "ntt"   # (the emulator cannot afford 2^13: the same path at 2^6 through the forced route)
    try:
        pc.lagrange_srs_beyond_2e12(6)
    finally:
        del os.environ["PLONK_LAGRANGE_SRS"]


def test_lookup_table_is_shared_across_contexts(emu):
    pc.lookup_table_is_shared_across_contexts()
    pc.lookup_table_colliding_key()


def test_lagrange_srs_paths(emu):
    from plonkathon_amd import Setup

    pc.lagrange_srs_paths(Setup.from_file(pc.PTAU))


def test_ntt_two_pass_wave_kernel(emu):
    """Two-pass transforms through the wave kernels' column and row passes, exact against the C oracle: 2^16 = 2^8 x 2^8
    (4 elements per thread in both passes), 2^17 = 2^8 x 2^9 (4, then 8) and forced to 2^9 x 2^8 (8, then 4: the column pass
    of the 8-element kernels takes its inter-pass twiddles from the two small tables).  Kernel kind 6: the throughput forms whatever
    the size of the call.  (The 512-thread 2^12 kernel as a ROW pass needs 2^19 points — a minute on the emulator: the GPU suite's
    2^24 = 2^12 x 2^12 case; as a single pass it is in test_ntt_forced_variants and test_ntt_extreme_inputs.)"""
    from oracle import c_oracle
    from plonkathon_amd import Basis, get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    check(ctx.L.plonk_ntt_select_kernel(ctx.handle, 6))  # (lone transforms of these sizes default to the latency forms: test_ntt_latency_forms)
    for log_n in (16, 17):
        v = pc.rand_vec(4000 + log_n, 1 << log_n)
        assert pc.ints(pc.P(v, Basis.MONOMIAL).fft()) == c_oracle.fr_ntt(v), log_n
    try:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, 17, 9))
        v = pc.rand_vec(4117, 1 << 17)
        assert pc.ints(pc.P(v, Basis.LAGRANGE).ifft()) == c_oracle.fr_ntt(v, True)
    finally:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, 17, 0))
    try:  # the inter-pass twiddles as two factors from the small tables (no budget for the full table), both directions
        check(ctx.L.plonk_ntt_set_table_budget(ctx.handle, 0))
        v = pc.rand_vec(4016, 1 << 16)
        assert pc.ints(pc.P(v, Basis.LAGRANGE).ifft()) == c_oracle.fr_ntt(v, True)
    finally:
        check(ctx.L.plonk_ntt_set_table_budget(ctx.handle, 4 << 30))
    assert pc.ints(pc.P(v, Basis.LAGRANGE).ifft()) == c_oracle.fr_ntt(v, True)  # ... and from the full table (1/N folded in)
    try:
        assert ctx.L.plonk_ntt_set_split(ctx.handle, 18, 14) != 0 and ctx.L.plonk_ntt_set_split(ctx.handle, 15, 9) != 0
    finally:
        check(ctx.L.plonk_ntt_set_split(ctx.handle, 18, 0))
        check(ctx.L.plonk_ntt_select_kernel(ctx.handle, 0))


def test_ntt_2_14_and_2_15(emu):
    pc.ntt_quad_sizes()


def test_bls12_381_ntt_2_14_and_2_15(emu):
    pc.bls_ntt_vs_oracle((14, 15), seed0=55, batch=2)


def test_bls12_381_golden_vectors(emu):
    pc.bls_golden(max_log_n=13)


def test_bls12_381_coset_transforms(emu):
    """n = 2^8 -> 4n = 2^10 (both E = 4), 2^9 -> 2^11 (both E = 8), 2^11 -> 2^13, 2^12 -> 2^14 (the four-point column path)."""
    pc.bls_coset_vs_oracle((8, 9), batch=2)
    pc.bls_coset_vs_oracle((11, 12), seed0=170)


def test_bls12_381_ntt_two_pass(emu):
    """2^16 = 2^8 x 2^8 over the BLS12-381 scalar field (inter-pass twiddles, 1/N folded into the hi table)."""
    pc.bls_ntt_vs_oracle((16,), seed0=91)


def test_async_upload(emu):
    from plonkathon_amd import Setup

    pc.async_upload_and_device_gather(Setup.from_file(pc.PTAU))


def test_g1_and_proof_encoding(emu):
    from plonkathon_amd import Setup

    pc.g1_encoding_cases(Setup.from_file(pc.PTAU))


def test_product_verifier(emu):
    from plonkathon_amd import Setup

    pc.verifier_cases(Setup.from_file(pc.PTAU))
