"""Device context and buffer ownership for the Python host layer.

One `Context` per (process, device) wraps a `plonk_ctx*`; `DeviceBuffer` owns one `plonk_mem_alloc`
allocation and frees it when garbage collected (SURVEY.md §8(b) "Ownership").
"""
import contextlib
import ctypes
import os

from . import _lib
from ._lib import check


class Context:
    def __init__(self, device=None):
        L = _lib.lib()
        if device is None:
            device = int(os.environ.get("PLONK_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = ctypes.c_int(0)
        check(L.plonk_device_count(ctypes.byref(n)))
        if n.value == 0:
            raise _lib.BackendError("no HIP device visible: plonkathon_amd needs an MI355X (there is no CPU fallback)")
        self.device = device % n.value
        self.handle = ctypes.c_void_p()
        check(L.plonk_ctx_create(self.device, ctypes.byref(self.handle)))
        self.L = L
        # Freed DeviceBuffers are kept for reuse, by size: hipFree waits for the device, and a reference-shaped proof
        # creates ~400 short-lived Polynomials (2.6 ms of hipFree per proof at group_order 2^11).  A context has ONE
        # stream, so a buffer's next user is ordered behind its last one — the same guarantee hipFreeAsync gives.
        self._pool, self._pool_bytes = {}, 0

    def name(self):
        buf = ctypes.create_string_buffer(256)
        check(self.L.plonk_ctx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def sync(self):
        check(self.L.plonk_ctx_sync(self.handle))

    def alloc(self, n_elems):
        return DeviceBuffer(self, n_elems)

    def upload_ints(self, ints):
        buf = DeviceBuffer(self, len(ints))
        if ints:
            check(self.L.plonk_fr_upload(self.handle, buf.ptr, b"".join(int(v).to_bytes(32, "little") for v in ints), len(ints)))
        return buf

    def upload_bytes(self, raw):
        """`raw` = canonical 32-byte little-endian elements back to back (the C-ABI's host format)."""
        assert len(raw) % 32 == 0
        buf = DeviceBuffer(self, len(raw) // 32)
        if raw:
            check(self.L.plonk_fr_upload(self.handle, buf.ptr, bytes(raw), len(raw) // 32))
        return buf

    def download_bytes(self, buf, n=None, offset=0):
        n = buf.n if n is None else n
        out = ctypes.create_string_buffer(32 * max(n, 1))
        if n:
            check(self.L.plonk_fr_download(self.handle, out, ctypes.c_void_p(buf.ptr.value + 32 * offset), n))
        return out.raw[: 32 * n]

    def download_ints(self, buf, n=None, offset=0):
        n = buf.n if n is None else n
        if n == 0:
            return []
        out = ctypes.create_string_buffer(32 * n)
        check(self.L.plonk_fr_download(self.handle, out, ctypes.c_void_p(buf.ptr.value + 32 * offset), n))
        raw = out.raw
        return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(n)]

    def host_alloc(self, nbytes):
        """A page-locked host buffer (hipHostMalloc) as a writable ctypes char array: what BatchProver.upload_values_async
        copies from without a host wait.  Freed with host_free (or with the process)."""
        ptr = ctypes.c_void_p()
        check(self.L.plonk_host_alloc(self.handle, nbytes, ctypes.byref(ptr)))
        buf = (ctypes.c_char * nbytes).from_address(ptr.value)
        buf._plonk_hptr = ptr
        return buf

    def host_free(self, buf):
        check(self.L.plonk_host_free(self.handle, buf._plonk_hptr))

    def mem_info(self):
        """(free, total) bytes of this context's device (hipMemGetInfo)."""
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        check(self.L.plonk_mem_info(self.handle, ctypes.byref(free), ctypes.byref(total)))
        return free.value, total.value

    def timer_start(self):
        check(self.L.plonk_timer_start(self.handle))

    def timer_stop_ms(self):
        ms = ctypes.c_float(0)
        check(self.L.plonk_timer_stop_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def msm_configure(self, window_bits=0, groups=0):
        """Bucket-method tuning knobs (0 = library default)."""
        check(self.L.plonk_msm_configure(self.handle, window_bits, groups))

    @contextlib.contextmanager
    def tuning(self, msm_window_bits=0, msm_groups=0, ntt_kind=0):
        """The diagnostics knobs of include/plonk_hip.h (plonk_msm_configure, plonk_ntt_select_kernel) for the length of a
        `with` block: they are per-context mutable state, and a caller that forgets to reset them measures the wrong thing
        afterwards — on exit the library defaults are back, whatever happened inside."""
        check(self.L.plonk_msm_configure(self.handle, msm_window_bits, msm_groups))
        if ntt_kind:
            check(self.L.plonk_ntt_select_kernel(self.handle, ntt_kind))
        try:
            yield self
        finally:
            if self.handle:
                self.L.plonk_msm_configure(self.handle, 0, 0)
                if ntt_kind:
                    self.L.plonk_ntt_select_kernel(self.handle, 0)

    def msm_lookup(self, mode=0, bits=0, budget_bytes=0, windows=False, top=False):
        """Table-MSM policy (include/plonk_hip.h): mode 0 auto, 1 off, 2 force `bits` for every base set; comb tables (bits =
        teeth; `top`: the comb of that many teeth with top tables) unless `windows` (bits = window bits: the layout of rounds
        2 - 5, kept for comparison)."""
        check(self.L.plonk_msm_lookup_configure(self.handle, mode | (16 if windows else 0) | (32 if top else 0), bits, budget_bytes))

    def profile(self, on):
        check(self.L.plonk_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        check(self.L.plonk_profile_reset(self.handle))

    def profile_read(self, kernel):
        """-> (total_ms, launches, algorithmic_bytes) recorded for `kernel` since the last reset."""
        ms, n, by = ctypes.c_double(0), ctypes.c_uint64(0), ctypes.c_double(0)
        check(self.L.plonk_profile_read(self.handle, kernel.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by)))
        return ms.value, n.value, by.value

    POOL_MAX_BYTES = 1 << 30       # retained in total
    POOL_MAX_BUFFER = 64 << 20     # larger buffers go straight back to the driver

    def _pool_take(self, nbytes):
        free = self._pool.get(nbytes)
        if free:
            self._pool_bytes -= nbytes
            return free.pop()
        return None

    def _pool_give(self, ptr, nbytes):
        if nbytes > self.POOL_MAX_BUFFER or self._pool_bytes + nbytes > self.POOL_MAX_BYTES:
            return False
        self._pool.setdefault(nbytes, []).append(ptr)
        self._pool_bytes += nbytes
        return True

    def trim(self):
        """Return every pooled buffer to the driver (call it before a large allocation elsewhere — an MSM table, a second
        context's twiddle tables — when HBM is nearly full: the C side does not know about this pool)."""
        for free in self._pool.values():
            for ptr in free:
                self.L.plonk_mem_free(self.handle, ptr)
        self._pool, self._pool_bytes = {}, 0

    def close(self):
        if self.handle:
            self.trim()
            self.L.plonk_ctx_destroy(self.handle)
            self.handle = None


class DeviceBuffer:
    """`n` Fr elements (32 B each, Montgomery form) in HBM."""

    def __init__(self, ctx, n_elems):
        self.ctx = ctx
        self.n = int(n_elems)
        self._nbytes = 32 * max(self.n, 1)
        self.ptr = ctx._pool_take(self._nbytes)
        if self.ptr is None:
            self.ptr = ctypes.c_void_p()
            rc = ctx.L.plonk_mem_alloc(ctx.handle, self._nbytes, ctypes.byref(self.ptr))
            if rc == _lib.PLONK_ERR_NOMEM and ctx._pool_bytes:
                # the pool may hold up to 1 GiB of freed buffers of OTHER sizes: hand them back to the driver and retry once
                ctx.trim()
                rc = ctx.L.plonk_mem_alloc(ctx.handle, self._nbytes, ctypes.byref(self.ptr))
            check(rc)

    def at(self, elem_offset):
        return ctypes.c_void_p(self.ptr.value + 32 * elem_offset)

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                if not self.ctx._pool_give(self.ptr, self._nbytes):
                    self.ctx.L.plonk_mem_free(self.ctx.handle, self.ptr)
                self.ptr = None
        except Exception:
            pass


class DeviceView:
    """`n` elements of another buffer starting at element `offset`: no storage of its own, keeps the parent alive."""

    def __init__(self, parent, offset, n_elems):
        self.ctx, self.parent, self.n = parent.ctx, parent, int(n_elems)
        self.ptr = ctypes.c_void_p(parent.ptr.value + 32 * offset)

    def at(self, elem_offset):
        return ctypes.c_void_p(self.ptr.value + 32 * elem_offset)


_default = None


def get_context():
    """The process-wide default context (created on first use; raises without a GPU)."""
    global _default
    if _default is None:
        _default = Context()
    return _default


def set_context(ctx):
    global _default
    _default = ctx
