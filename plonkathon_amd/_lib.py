"""ctypes binding of libplonk_hip.so (the C-ABI in include/plonk_hip.h).

The HIP library is the only compute path: if it is missing, not loadable, or no GPU is visible,
importing/using the backend raises — there is no CPU fallback in this package.  (The CPU test-suite
injects a host build of the same kernel sources through `bind()`; see tests/emu/.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLONK_HIP_LIB: path of an alternative build of the same library (tuning experiments)
LIB_PATH = os.environ.get("PLONK_HIP_LIB") or os.path.join(_HERE, "libplonk_hip.so")

c_void_pp = ctypes.POINTER(ctypes.c_void_p)
_u8p = ctypes.c_char_p

# name -> (restype, argtypes); mirrors include/plonk_hip.h one to one
SIGNATURES = {
    "plonk_last_error": (ctypes.c_char_p, []),
    "plonk_abi_version": (ctypes.c_int, []),
    "plonk_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "plonk_ctx_create": (ctypes.c_int, [ctypes.c_int, c_void_pp]),
    "plonk_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_ctx_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_ctx_device_name": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    "plonk_mem_alloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]),
    "plonk_mem_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "plonk_mem_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "plonk_mem_h2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_mem_d2h": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_mem_d2d": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_mem_zero": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_fr_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_size_t]),
    "plonk_fr_download": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_fr_ntt": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_size_t]),
    "plonk_ntt_set_table_budget": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_bls_fr_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _u8p, ctypes.c_size_t]),
    "plonk_bls_fr_download": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_bls_fr_ntt": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_size_t]),
    "plonk_bls_fr_coset_extend": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_size_t]),
    "plonk_bls_fr_coset_to_coeffs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_size_t]),
    "plonk_ntt_configure": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]),
    "plonk_fr_powers": (ctypes.c_int, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_fr_equal": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]),
    "plonk_fr_grand_product": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_void_p] * 6 + [ctypes.c_uint, _u8p, _u8p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]),
    "plonk_fr_quotient": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_void_p), _u8p, _u8p, _u8p, _u8p, ctypes.c_void_p]),
    "plonk_g1_compress": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, ctypes.c_char_p]),
    "plonk_g1_decompress": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p]),
    "plonk_prover_download_compressed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p]),
    "plonk_host_alloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]),
    "plonk_host_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "plonk_prover_upload_variables_async": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_gather_proofs_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]),
    "plonk_ntt_select_kernel": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint]),
    "plonk_ntt_set_split": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]),
    "plonk_ntt_get_split": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)]),
    "plonk_fr_coset_extend": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_size_t]),
    "plonk_fr_coset_to_coeffs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_size_t]),
    "plonk_fr_coset_ntt_from_coeffs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, _u8p, ctypes.c_size_t]),
    "plonk_fr_pointwise": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_fr_scalar_op": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, _u8p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]),
    "plonk_fr_rotate": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]),
    "plonk_fr_batch_inverse": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_fr_barycentric": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_void_p]),
    "plonk_fr_barycentric_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_char_p, ctypes.c_char_p]),
    "plonk_fr_lincomb": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_srs_load_ptau": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, c_void_pp]),
    "plonk_srs_load_affine": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, c_void_pp]),
    "plonk_srs_lagrange": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, c_void_pp]),
    "plonk_srs_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "plonk_srs_size": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]),
    "plonk_srs_lookup_bits": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]),
    "plonk_srs_lookup_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "plonk_srs_lookup_layout": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    "plonk_srs_lookup_top": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    "plonk_msm_lookup_configure": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t]),
    "plonk_g1_msm": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "plonk_msm_configure": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]),
    "plonk_prover_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, _u8p, ctypes.c_size_t, c_void_pp]),
    "plonk_prover_set_options": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint]),
    "plonk_prover_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_prover_upload_witness": (ctypes.c_int, [ctypes.c_void_p, _u8p, _u8p, ctypes.c_size_t]),
    "plonk_prover_set_wiring": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_prover_upload_variables": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_prover_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_prover_download": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "plonk_prover_challenges": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, c_void_pp]),
    "plonk_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_comm_size": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "plonk_gather_results": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_comm_max_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    "plonk_comm_barrier": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_comm_set_default_timeout": (ctypes.c_int, [ctypes.c_double]),
    "plonk_comm_set_timeout": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    "plonk_device_peer_access": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_size_t]),
    "plonk_comm_last_gather_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "plonk_comm_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
    "plonk_fr_ntt_dist_columns": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int]),
    "plonk_fr_ntt_dist_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int]),
    "plonk_comm_all_to_all": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_fr_ntt_distributed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]),
    "plonk_pairing_check": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]),
    "plonk_transcript_new": (ctypes.c_int, [_u8p, ctypes.c_size_t, c_void_pp]),
    "plonk_transcript_clone": (ctypes.c_int, [ctypes.c_void_p, c_void_pp]),
    "plonk_transcript_free": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_transcript_append_message": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, _u8p, ctypes.c_size_t]),
    "plonk_transcript_challenge_bytes": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "plonk_transcript_challenge_scalar": (ctypes.c_int, [ctypes.c_void_p, _u8p, ctypes.c_size_t, ctypes.c_void_p]),
    "plonk_profile_enable": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "plonk_profile_read": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]),
    "plonk_profile_reset": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_timer_start": (ctypes.c_int, [ctypes.c_void_p]),
    "plonk_timer_stop_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
}

PLONK_OK, PLONK_ERR_ARG, PLONK_ERR_HIP, PLONK_ERR_NOMEM, PLONK_ERR_STATE, PLONK_ERR_TIMEOUT = 0, -1, -2, -3, -4, -5
OP_ADD, OP_SUB, OP_MUL, OP_DIV = 0, 1, 2, 3


class BackendError(RuntimeError):
    """A HIP-level failure inside libplonk_hip.so."""


_lib = None


def bind(cdll):
    """Attach prototypes to an already-opened library exporting the plonk_* C-ABI."""
    global _lib
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError here == ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = cdll
    return cdll


def lib():
    """The loaded HIP library; raises if it cannot be loaded (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendError(
                "libplonk_hip.so not found at %s — build it with `make -C plonkathon_amd/csrc` "
                "(or __graft_entry__.build()); plonkathon_amd has no CPU fallback" % LIB_PATH
            )
        bind(ctypes.CDLL(LIB_PATH))
    return _lib


def check(rc):
    """Maps a C-ABI status to the exception the reference would raise at that point."""
    if rc == PLONK_OK:
        return
    msg = lib().plonk_last_error().decode("utf-8", "replace")
    if rc == PLONK_ERR_ARG:
        raise AssertionError(msg)  # the reference guards these conditions with `assert`
    if rc == PLONK_ERR_NOMEM:
        raise MemoryError(msg)
    if rc == PLONK_ERR_TIMEOUT:
        raise TimeoutError(msg)  # a collective's deadline passed: another rank is dead or stuck (plonk_comm_set_timeout)
    raise BackendError("libplonk_hip: %s (status %d)" % (msg, rc))
