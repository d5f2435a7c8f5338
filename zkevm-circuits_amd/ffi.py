"""ctypes binding of libzkhip.so (include/zkhip.h).  Plumbing only: every computation is a HIP kernel.

Field elements travel as numpy uint64 arrays whose last axis holds the 4 Montgomery limbs (8 for affine
points, 12 for Jacobian), exactly the in-memory layout of `halo2curves::bn256::{Fr, G1Affine, G1}`.
"""
from __future__ import annotations

import bisect
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkhip.so")
_LIB = None

EXPORTS = [
    "zkhip_create", "zkhip_destroy", "zkhip_last_error", "zkhip_version", "zkhip_set_stream", "zkhip_sync",
    "zkhip_dev_alloc", "zkhip_dev_free", "zkhip_mem_info", "zkhip_upload", "zkhip_download", "zkhip_copy_dev",
    "zkhip_host_alloc", "zkhip_host_free", "zkhip_upload_async", "zkhip_upload_overlapped", "zkhip_upload_fence",
    "zkhip_field_binop", "zkhip_field_batch_invert", "zkhip_g1_add_affine", "zkhip_g1_multiples", "zkhip_probe_mad_rate", "zkhip_selftest_arithmetic",
    "zkhip_register_bases", "zkhip_release_bases", "zkhip_msm", "zkhip_msm_batch_dev", "zkhip_msm_range_dev",
    "zkhip_g1_sum_jacobian", "zkhip_g1_to_affine", "zkhip_g1_fft",
    "zkhip_ntt", "zkhip_ntt_dev",
    "zkhip_domain_create", "zkhip_domain_destroy", "zkhip_domain_k", "zkhip_domain_extended_k",
    "zkhip_domain_quotient_poly_degree", "zkhip_domain_constant",
    "zkhip_lagrange_to_coeff", "zkhip_coeff_to_lagrange", "zkhip_coeff_to_extended", "zkhip_extended_to_coeff",
    "zkhip_divide_by_vanishing_poly",
    "zkhip_lagrange_to_coeff_dev", "zkhip_coeff_to_lagrange_dev", "zkhip_coeff_to_extended_dev",
    "zkhip_extended_to_coeff_dev", "zkhip_divide_by_vanishing_poly_dev",
    "zkhip_coeff_to_extended_split_dev", "zkhip_extend_half_dev", "zkhip_coeff_to_extended_half_dev",
    "zkhip_evaluate_h_columns_used", "zkhip_evaluate_graphs_dev",
    "zkhip_evaluate_h", "zkhip_evaluate_h_last_stats", "zkhip_evaluate_h_cache_stats", "zkhip_evaluate_h_prepare",
    "zkhip_coeff_to_cosets_dev", "zkhip_extended_to_cosets_dev", "zkhip_cosets_to_extended_dev",
    "zkhip_batch_invert_dev", "zkhip_prefix_product_dev", "zkhip_prefix_product_noalloc_dev", "zkhip_permutation_product_dev", "zkhip_lookup_product_dev", "zkhip_lookup_product_batch_dev",
    "zkhip_eval_polynomial_dev", "zkhip_eval_polynomials_dev", "zkhip_poly_lincomb_dev", "zkhip_kate_division_dev", "zkhip_lookup_permute_dev", "zkhip_lookup_permute_batch_dev", "zkhip_permutation_sigma_dev",
    "zkhip_timing_enable", "zkhip_timing_reset", "zkhip_timing_get",
]


class ZkhipError(RuntimeError):
    pass


def lib():
    """Loads libzkhip.so.  There is no fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ZkhipError(f"{LIB_PATH} is missing: build it with `python -m` __graft_entry__.build() "
                             "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _LIB = C.CDLL(LIB_PATH)
        _LIB.zkhip_last_error.restype = C.c_char_p
        _LIB.zkhip_version.restype = C.c_char_p
        for name in ("zkhip_domain_k", "zkhip_domain_extended_k", "zkhip_domain_quotient_poly_degree"):
            getattr(_LIB, name).restype = C.c_uint32
        _LIB.zkhip_destroy.restype = None
        _LIB.zkhip_domain_destroy.restype = None
    return _LIB


# ----------------------------------------------------------------------------- descriptor structs
class Graph(C.Structure):
    _fields_ = [("constants", C.c_void_p), ("num_constants", C.c_uint32),
                ("rotations", C.c_void_p), ("num_rotations", C.c_uint32),
                ("calc_words", C.c_void_p), ("num_calc_words", C.c_uint32),
                ("num_calculations", C.c_uint32)]


class Lookup(C.Structure):
    _fields_ = [("graph", Graph), ("product", C.c_void_p), ("permuted_input", C.c_void_p),
                ("permuted_table", C.c_void_p)]


class EvalhDesc(C.Structure):
    _fields_ = [("mem", C.c_uint32), ("poly_form", C.c_uint32),
                ("num_fixed", C.c_uint32), ("num_advice", C.c_uint32), ("num_instance", C.c_uint32),
                ("num_challenges", C.c_uint32),
                ("fixed_cosets", C.POINTER(C.c_void_p)), ("advice", C.POINTER(C.c_void_p)),
                ("instance", C.POINTER(C.c_void_p)), ("challenges", C.c_void_p),
                ("y", C.c_uint64 * 4), ("beta", C.c_uint64 * 4), ("gamma", C.c_uint64 * 4), ("theta", C.c_uint64 * 4),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("custom_gates", C.POINTER(Graph)),
                ("num_lookups", C.c_uint32), ("lookups", C.POINTER(Lookup)),
                ("num_perm_columns", C.c_uint32), ("perm_column_kind", C.c_void_p),
                ("perm_column_index", C.c_void_p), ("perm_sigma_cosets", C.POINTER(C.c_void_p)),
                ("num_perm_sets", C.c_uint32), ("perm_product_cosets", C.POINTER(C.c_void_p)),
                ("perm_chunk_len", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("num_cosets", C.c_uint32), ("cosets", C.c_void_p), ("basis", C.c_uint32),
                ("row_first", C.c_uint32), ("row_count", C.c_uint32),
                ("term_select", C.c_uint32), ("term_degree_max", C.c_uint32)]


MEM_HOST, MEM_DEVICE = 0, 1
FORM_COEFF, FORM_EXTENDED = 0, 1
BASIS_EXTENDED, BASIS_LAGRANGE = 0, 1
TERMS_ALL, TERMS_LOW, TERMS_HIGH = 0, 1, 2


def _p(a):
    return C.c_void_p(a.ctypes.data)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class DescBuilder:
    """Builds a zkhip_evalh_desc from numpy / device pointers and keeps every referenced buffer alive."""

    def __init__(self):
        self.keep = []

    def graph(self, g, to_mont) -> Graph:
        """g: plonk.GraphEvaluator; to_mont: canonical int -> 4 Montgomery limbs.  The serialised arrays are kept on the
        graph (until it grows): a proving key's graphs are serialised once, not per call."""
        key = (len(g.constants), len(g.rotations), len(g.calculations))
        cached = g.__dict__.get("_wire_cache")
        if cached is None or cached[0] != key:
            cached = (key, np.array([to_mont(c) for c in g.constants], dtype=np.uint64).reshape(-1, 4),
                      np.array(g.rotations, dtype=np.int32), np.array(g.wire_words(), dtype=np.uint32))
            g.__dict__["_wire_cache"] = cached
        _, consts, rots, words = cached
        self.keep += [consts, rots, words]
        return Graph(_p(consts), len(g.constants), _p(rots) if len(rots) else None, len(rots),
                     _p(words), len(words), len(g.calculations))

    def ptr_array(self, ptrs):
        arr = (C.c_void_p * max(len(ptrs), 1))(*[C.c_void_p(int(p)) for p in ptrs])
        self.keep.append(arr)
        return C.cast(arr, C.POINTER(C.c_void_p))

    def host_ptrs(self, arrays):
        arrays = [u64(a) for a in arrays]
        self.keep += arrays
        return self.ptr_array([a.ctypes.data for a in arrays])

    def limbs(self, v4):
        return (C.c_uint64 * 4)(*[int(x) for x in v4])


# ----------------------------------------------------------------------------- context
class Context:
    def __init__(self, device: int = 0):
        self._l = lib()
        h = C.c_void_p()
        rc = self._l.zkhip_create(C.c_int(device), C.byref(h))
        if rc != 0:
            raise ZkhipError(f"zkhip_create(device={device}) failed with status {rc}: no usable gfx950 device "
                             "(there is no CPU fallback)")
        self.h = h
        self.device = device
        # bytes handed out by alloc() and not yet freed, and their high-water mark: what a rank keeps resident
        self._sizes = {}
        self._pinned = {}                 # address -> bytes of the page-locked host buffers handed out by host_zeros
        self._pinned_index = None         # sorted (base, end) pairs of the above, rebuilt after a change
        self._stage = None                # page-locked ring for small uploads (see upload)
        self._stage_at = 0
        self._stage_small = os.environ.get("ZKHIP_STAGE_SMALL", "1") != "0"
        self.allocated_bytes = 0
        self.peak_allocated_bytes = 0
        # Freed blocks are kept by size and handed out again: a proof allocates and frees well over a hundred column-sized
        # buffers, and every hipFree waits for the device.  Reuse is safe on the context's stream order (the library's
        # internal streams join it at the end of each call; overlapped uploads start behind it).
        self._pool = {}
        self._pooled_bytes = 0
        self._upload_window = False       # overlapped copies issued since the last fence

    POOL_CAP_BYTES = 4 << 30

    def release_pool(self):
        """returns every pooled block to the device"""
        for size, blocks in self._pool.items():
            for d in blocks:
                self._l.zkhip_dev_free(self.h, C.c_void_p(d))
        self._pool = {}
        self._pooled_bytes = 0

    def close(self):
        if getattr(self, "h", None):
            self.release_pool()
            for addr in list(getattr(self, "_pinned", {})):
                self._l.zkhip_host_free(self.h, C.c_void_p(addr))
            self._pinned = {}
            self._l.zkhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise ZkhipError(f"status {rc}: {self._l.zkhip_last_error(self.h).decode()}")

    # -- memory ---------------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._ck(self._l.zkhip_set_stream(self.h, C.c_void_p(stream_handle)))

    def sync(self):
        self._ck(self._l.zkhip_sync(self.h))

    def alloc(self, nbytes: int) -> int:
        blocks = self._pool.get(nbytes)
        if blocks:
            addr = blocks.pop()
            self._pooled_bytes -= nbytes
        else:
            d = C.c_void_p()
            rc = self._l.zkhip_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(d))
            if rc != 0 and self._pooled_bytes:        # the pool may hold what this request needs
                self.release_pool()
                rc = self._l.zkhip_dev_alloc(self.h, C.c_size_t(nbytes), C.byref(d))
            self._ck(rc)
            addr = d.value
        self._sizes[addr] = nbytes
        self.allocated_bytes += nbytes
        self.peak_allocated_bytes = max(self.peak_allocated_bytes, self.allocated_bytes)
        return addr

    def free(self, dptr: int):
        if self._upload_window:
            # Copies on the transfer stream are ordered behind the compute stream only at the first copy after a fence.  A
            # block freed now may be handed out again and become the target of a later copy of the same window, which
            # would not wait for the kernels that still read the block: the window is closed first.
            self.upload_fence()
        size = self._sizes.pop(dptr, None)
        if size is None:                              # not one of alloc()'s blocks
            self._ck(self._l.zkhip_dev_free(self.h, C.c_void_p(dptr)))
            return
        self.allocated_bytes -= size
        if self._pooled_bytes + size <= self.POOL_CAP_BYTES:
            self._pool.setdefault(size, []).append(dptr)
            self._pooled_bytes += size
        else:
            self._ck(self._l.zkhip_dev_free(self.h, C.c_void_p(dptr)))

    def mem_info(self):
        """(free, total) bytes of device memory.  Blocks waiting in the pool (at most POOL_CAP_BYTES) are NOT counted as
        free: `alloc` returns them to the device when one of its own requests needs the room, but the library's internal
        buffers (transform scratch, MSM workspaces, the evaluate_h program) are allocated past the pool, so a caller that
        sizes its working set from this figure must not be promised the pooled bytes."""
        free, total = C.c_size_t(), C.c_size_t()
        self._ck(self._l.zkhip_mem_info(self.h, C.byref(free), C.byref(total)))
        return free.value, total.value

    def reset_peak(self):
        self.peak_allocated_bytes = self.allocated_bytes

    STAGE_BYTES = 8 << 20          # ring of page-locked memory for small uploads
    STAGE_MAX = 256 << 10          # arrays up to this size go through the ring

    def upload(self, arr, dptr: int | None = None) -> int:
        arr = u64(arr)
        if dptr is None:
            dptr = self.alloc(arr.nbytes)
        if self.is_pinned(arr):
            # page-locked source: a DMA ordered on the context's stream, no wait (the buffer outlives the context's work)
            self._ck(self._l.zkhip_upload_async(self.h, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes)))
        elif 0 < arr.nbytes <= self.STAGE_MAX and self._stage_small:
            # small arrays in ordinary memory (blinding rows, scalars of a phase): copied into a page-locked ring and sent
            # from there without waiting.  A synchronous copy would make the host wait for everything queued before it
            # -- two hundred times in a proof, with the device idle while the host catches up afterwards.
            if self._stage is None:
                self._stage = self.host_zeros((self.STAGE_BYTES // 8,))
                self._stage_at = 0
            words = (arr.size + 31) // 32 * 32                     # slots start on 256-byte boundaries
            if self._stage_at + words > self._stage.size:
                self.sync()                                        # every copy out of the ring has been made
                self._stage_at = 0
            view = self._stage[self._stage_at:self._stage_at + arr.size]
            view[...] = arr.reshape(-1)
            self._stage_at += words
            self._ck(self._l.zkhip_upload_async(self.h, C.c_void_p(dptr), _p(view), C.c_size_t(arr.nbytes)))
        else:
            self._ck(self._l.zkhip_upload(self.h, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes)))
        return dptr

    def upload_overlapped(self, arr: np.ndarray, dptr: int):
        """a copy from page-locked memory on the context's transfer stream: it runs beside the calls that follow, which
        must not touch the target before `upload_fence`"""
        if arr.dtype != np.uint64 or not arr.flags.c_contiguous or not self.is_pinned(arr):
            raise ValueError("upload_overlapped takes a contiguous uint64 array in page-locked memory (host_zeros)")
        self._ck(self._l.zkhip_upload_overlapped(self.h, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes)))
        self._upload_window = True

    def upload_fence(self):
        """the calls that follow see every overlapped copy issued so far"""
        self._ck(self._l.zkhip_upload_fence(self.h))
        self._upload_window = False

    def host_zeros(self, shape) -> np.ndarray:
        """a zeroed uint64 array in page-locked host memory (freed with the context): witness columns built in such
        arrays reach the device by DMA at the link's rate, without the staging copy ordinary memory needs"""
        nbytes = int(np.prod(shape)) * 8
        h = C.c_void_p()
        self._ck(self._l.zkhip_host_alloc(self.h, C.c_size_t(max(nbytes, 8)), C.byref(h)))
        self._pinned[h.value] = max(nbytes, 8)
        self._pinned_index = None
        buf = (C.c_uint8 * max(nbytes, 8)).from_address(h.value)
        a = np.frombuffer(buf, dtype=np.uint64, count=nbytes // 8).reshape(shape)
        a[...] = 0
        return a

    def host_free(self, arr: np.ndarray):
        addr = arr.ctypes.data
        if addr in self._pinned:
            self._ck(self._l.zkhip_host_free(self.h, C.c_void_p(addr)))
            del self._pinned[addr]
            self._pinned_index = None

    def is_pinned(self, arr: np.ndarray) -> bool:
        if self._pinned_index is None:
            self._pinned_index = sorted((base, base + size) for base, size in self._pinned.items())
        addr = arr.ctypes.data
        i = bisect.bisect_right(self._pinned_index, (addr, float("inf"))) - 1
        return i >= 0 and addr + arr.nbytes <= self._pinned_index[i][1]

    def copy_dev(self, d_dst: int, d_src: int, nbytes: int):
        """device-to-device copy on the context's stream (asynchronous)"""
        self._ck(self._l.zkhip_copy_dev(self.h, C.c_void_p(d_dst), C.c_void_p(d_src), C.c_size_t(nbytes)))

    def download(self, dptr: int, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.uint64)
        self._ck(self._l.zkhip_download(self.h, _p(out), C.c_void_p(dptr), C.c_size_t(out.nbytes)))
        return out

    # -- field / curve --------------------------------------------------------------------------
    def field_binop(self, field: int, op: int, a, b):
        a, b = u64(a), u64(b)
        out = np.empty_like(a)
        self._ck(self._l.zkhip_field_binop(self.h, C.c_int(field), C.c_int(op), _p(a), _p(b), _p(out),
                                           C.c_size_t(a.size // 4)))
        return out

    def field_batch_invert(self, field: int, a):
        a = u64(a)
        out = np.empty_like(a)
        self._ck(self._l.zkhip_field_batch_invert(self.h, C.c_int(field), _p(a), _p(out), C.c_size_t(a.size // 4)))
        return out

    def g1_add_affine(self, p, q):
        p, q = u64(p), u64(q)
        out = np.empty_like(p)
        self._ck(self._l.zkhip_g1_add_affine(self.h, _p(p), _p(q), _p(out), C.c_size_t(p.size // 8)))
        return out

    def g1_multiples(self, n: int, first: int = 1, base=None):
        """[(first + i) * base] for i < n, affine; base defaults to the generator (1, 2)."""
        if base is None:
            one = [0xd35d438dc58f0d9d, 0x0a78eb28f5c70b3d, 0x666ea36f7879462c, 0x0e0a77c19a07df2f]   # Fq one (Montgomery)
            two = [0xa6ba871b8b1e1b3a, 0x14f1d651eb8e167b, 0xccdd46def0f28c58, 0x1c14ef83340fbe5e]   # Fq two
            base = np.array(one + two, dtype=np.uint64)
        base = u64(base)
        out = np.empty((n, 8), dtype=np.uint64)
        self._ck(self._l.zkhip_g1_multiples(self.h, _p(base), C.c_uint64(first), C.c_size_t(n), _p(out)))
        return out

    def g1_fft(self, affine_xy, log_n: int, omega, scale=None):
        """best_fft over G1 points: (2^log_n, 8) affine limbs in, same shape out"""
        pts = u64(affine_xy)
        assert pts.size == 8 << log_n
        out = np.empty((1 << log_n, 8), dtype=np.uint64)
        om = u64(omega)
        sc = u64(scale) if scale is not None else None
        self._ck(self._l.zkhip_g1_fft(self.h, _p(pts), C.c_uint32(log_n), _p(om), _p(sc) if sc is not None else None, _p(out)))
        return out

    def selftest_arithmetic(self, seed=1, quads=1 << 16):
        """bit masks of failed checks of the lazy device arithmetic (Fr, Fr chained, Fq, Fq chained); all zero = pass"""
        out = (C.c_uint32 * 4)()
        self._ck(self._l.zkhip_selftest_arithmetic(self.h, C.c_uint64(seed), C.c_size_t(quads), out))
        return [int(v) for v in out]

    def probe_mad_rate(self, iters=2048):
        mads, muls = C.c_double(), C.c_double()
        self._ck(self._l.zkhip_probe_mad_rate(self.h, C.c_int(iters), C.byref(mads), C.byref(muls)))
        return mads.value, muls.value

    # -- MSM ------------------------------------------------------------------------------------
    def register_bases(self, affine_xy) -> int:
        a = u64(affine_xy)
        bid = C.c_uint32()
        self._ck(self._l.zkhip_register_bases(self.h, _p(a), C.c_size_t(a.size // 8), C.byref(bid)))
        return bid.value

    def release_bases(self, bid: int):
        self._ck(self._l.zkhip_release_bases(self.h, C.c_uint32(bid)))

    def msm(self, bid: int, scalars):
        """`best_multiexp(coeffs, bases[..n])` -> Jacobian limbs (12,)."""
        s = u64(scalars)
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self._l.zkhip_msm(self.h, C.c_uint32(bid), _p(s), C.c_size_t(s.size // 4), _p(out)))
        return out

    def msm_batch_dev(self, bid: int, d_scalars, n: int):
        ptrs = (C.c_void_p * len(d_scalars))(*[C.c_void_p(int(p)) for p in d_scalars])
        out = np.zeros((len(d_scalars), 12), dtype=np.uint64)
        self._ck(self._l.zkhip_msm_batch_dev(self.h, C.c_uint32(bid), ptrs, C.c_size_t(len(d_scalars)),
                                             C.c_size_t(n), _p(out)))
        return out

    def msm_range_dev(self, bid: int, first: int, d_scalars: int, n: int):
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self._l.zkhip_msm_range_dev(self.h, C.c_uint32(bid), C.c_size_t(first), C.c_void_p(d_scalars),
                                             C.c_size_t(n), _p(out)))
        return out

    def g1_sum_jacobian(self, jacs):
        j = u64(jacs)
        out = np.zeros(12, dtype=np.uint64)
        self._ck(self._l.zkhip_g1_sum_jacobian(_p(j), C.c_size_t(j.size // 12), _p(out)))
        return out

    def g1_to_affine(self, jac):
        j = u64(jac)
        out = np.zeros(8, dtype=np.uint64)
        self._ck(self._l.zkhip_g1_to_affine(_p(j), _p(out)))
        return out

    # -- NTT ------------------------------------------------------------------------------------
    def ntt(self, a, log_n: int, omega):
        """`best_fft(a, omega, log_n)`; returns the transformed copy."""
        a = u64(a).copy()
        om = u64(omega)
        self._ck(self._l.zkhip_ntt(self.h, _p(a), C.c_uint32(log_n), _p(om)))
        return a

    def ntt_dev(self, d_a: int, log_n: int, omega, batch: int = 1):
        om = u64(omega)
        self._ck(self._l.zkhip_ntt_dev(self.h, C.c_void_p(d_a), C.c_uint32(log_n), _p(om), C.c_size_t(batch)))

    # -- timing ---------------------------------------------------------------------------------
    # grand products and openings (device-resident)
    def batch_invert_dev(self, field: int, d_a: int, n: int):
        self._ck(self._l.zkhip_batch_invert_dev(self.h, C.c_int(field), C.c_void_p(d_a), C.c_size_t(n)))

    def prefix_product_dev(self, d_m: int, d_out: int, n: int, init):
        init = u64(init)
        self._ck(self._l.zkhip_prefix_product_dev(self.h, C.c_void_p(d_m), C.c_void_p(d_out), C.c_size_t(n), _p(init)))

    def prefix_product_noalloc_dev(self, d_m: int, d_out: int, n: int, init):
        """prefix_product_dev over the library's static block-total buffer: no device allocation, n <= 2^24"""
        init = u64(init)
        self._ck(self._l.zkhip_prefix_product_noalloc_dev(self.h, C.c_void_p(d_m), C.c_void_p(d_out), C.c_size_t(n), _p(init)))

    def eval_polynomial_dev(self, d_coeff: int, n: int, x):
        x = u64(x)
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self._l.zkhip_eval_polynomial_dev(self.h, C.c_void_p(d_coeff), C.c_size_t(n), _p(x), _p(out)))
        return out

    def eval_polynomials_dev(self, d_coeffs, points, n: int):
        """out[i] = polynomial d_coeffs[i] (n coefficients, device) at points[i] ((count, 4) Montgomery limbs)"""
        count = len(d_coeffs)
        ptrs = (C.c_void_p * max(count, 1))(*[C.c_void_p(int(p)) for p in d_coeffs])
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint64).reshape(count, 4)) if count else np.zeros((1, 4), np.uint64)
        out = np.zeros((max(count, 1), 4), dtype=np.uint64)
        self._ck(self._l.zkhip_eval_polynomials_dev(self.h, ptrs, _p(pts), C.c_uint32(count), C.c_size_t(n), _p(out)))
        return out[:count]

    def poly_lincomb_dev(self, d_polys, lens, coeffs, n: int, d_out: int):
        """d_out[i] = sum_k coeffs[k] * d_polys[k][i]; d_polys: device pointers, lens: their lengths (<= n),
        coeffs: (num, 4) Montgomery limbs."""
        num = len(d_polys)
        ptrs = (C.c_void_p * max(num, 1))(*[C.c_void_p(int(p)) for p in d_polys])
        ls = (C.c_size_t * max(num, 1))(*[int(x) for x in lens])
        cf = np.ascontiguousarray(np.asarray(coeffs, dtype=np.uint64).reshape(num, 4)) if num else np.zeros((1, 4), np.uint64)
        self._ck(self._l.zkhip_poly_lincomb_dev(self.h, ptrs, ls, _p(cf), C.c_uint32(num), C.c_size_t(n), C.c_void_p(d_out)))

    def kate_division_dev(self, d_poly: int, n: int, point, d_out: int):
        point = u64(point)
        self._ck(self._l.zkhip_kate_division_dev(self.h, C.c_void_p(d_poly), C.c_size_t(n), _p(point), C.c_void_p(d_out)))

    def evaluate_h_last_stats(self):
        out = np.zeros(4, dtype=np.uint64)
        self._ck(self._l.zkhip_evaluate_h_last_stats(self.h, _p(out)))
        return {"instructions": int(out[0]), "multiplications": int(out[1]), "column_reads": int(out[2]), "slots": int(out[3])}

    def evaluate_h_cache_stats(self):
        out = np.zeros(2, dtype=np.uint64)
        self._ck(self._l.zkhip_evaluate_h_cache_stats(self.h, _p(out)))
        return {"builds": int(out[0]), "hits": int(out[1])}

    def timing_enable(self, on=True):
        """on: False / True, or 2 = only the dominant kernels ("msm_accumulate", "ntt"), which disturbs a measured step least"""
        self._ck(self._l.zkhip_timing_enable(self.h, C.c_int(2 if on == 2 and on is not True else (1 if on else 0))))

    def timing_reset(self):
        self._ck(self._l.zkhip_timing_reset(self.h))

    def timing_get(self, family: str):
        us, cnt = C.c_double(), C.c_uint64()
        self._ck(self._l.zkhip_timing_get(self.h, family.encode(), C.byref(us), C.byref(cnt)))
        return us.value, cnt.value


class Domain:
    """`EvaluationDomain::new(j, k)` on the device."""

    def __init__(self, ctx: Context, j: int, k: int):
        self.ctx = ctx
        self._l = ctx._l
        h = C.c_void_p()
        ctx._ck(self._l.zkhip_domain_create(ctx.h, C.c_uint32(j), C.c_uint32(k), C.byref(h)))
        self.h = h
        self.k = self._l.zkhip_domain_k(h)
        self.extended_k = self._l.zkhip_domain_extended_k(h)
        self.quotient_poly_degree = self._l.zkhip_domain_quotient_poly_degree(h)
        self.n = 1 << self.k
        self.extended_len = 1 << self.extended_k

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self._l.zkhip_domain_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def constant(self, which: int):
        out = np.zeros(4, dtype=np.uint64)
        self.ctx._ck(self._l.zkhip_domain_constant(self.h, C.c_int(which), _p(out)))
        return out

    def lagrange_to_coeff(self, a):
        a = u64(a).copy()
        assert a.size // 4 == self.n
        self.ctx._ck(self._l.zkhip_lagrange_to_coeff(self.h, _p(a)))
        return a

    def coeff_to_lagrange(self, a):
        a = u64(a).copy()
        assert a.size // 4 == self.n
        self.ctx._ck(self._l.zkhip_coeff_to_lagrange(self.h, _p(a)))
        return a

    def coeff_to_extended(self, a):
        a = u64(a)
        assert a.size // 4 == self.n
        out = np.empty((self.extended_len, 4), dtype=np.uint64)
        self.ctx._ck(self._l.zkhip_coeff_to_extended(self.h, _p(a), _p(out)))
        return out

    def extended_to_coeff(self, a):
        a = u64(a)
        assert a.size // 4 == self.extended_len
        out = np.empty((self.n * self.quotient_poly_degree, 4), dtype=np.uint64)
        self.ctx._ck(self._l.zkhip_extended_to_coeff(self.h, _p(a), _p(out)))
        return out

    def divide_by_vanishing_poly(self, a):
        a = u64(a).copy()
        assert a.size // 4 == self.extended_len
        self.ctx._ck(self._l.zkhip_divide_by_vanishing_poly(self.h, _p(a)))
        return a

    # device-resident, batched
    def lagrange_to_coeff_dev(self, d_a: int, batch: int = 1):
        self.ctx._ck(self._l.zkhip_lagrange_to_coeff_dev(self.h, C.c_void_p(d_a), C.c_size_t(batch)))

    def coeff_to_lagrange_dev(self, d_a: int, batch: int = 1):
        self.ctx._ck(self._l.zkhip_coeff_to_lagrange_dev(self.h, C.c_void_p(d_a), C.c_size_t(batch)))

    def coeff_to_extended_dev(self, d_coeff: int, d_out: int, batch: int = 1):
        self.ctx._ck(self._l.zkhip_coeff_to_extended_dev(self.h, C.c_void_p(d_coeff), C.c_void_p(d_out),
                                                         C.c_size_t(batch)))

    def coeff_to_extended_split_dev(self, d_coeff: int, d_out: int, batch: int = 1):
        """coeff_to_extended with each result split by row parity: even-indexed extended rows, then the odd-indexed ones
        (the compact layouts of the even and of the odd cosets)"""
        self.ctx._ck(self._l.zkhip_coeff_to_extended_split_dev(self.h, C.c_void_p(d_coeff), C.c_void_p(d_out),
                                                               C.c_size_t(batch)))

    def extend_half_dev(self, d_even: int, d_odd: int, batch: int = 1):
        """values of a polynomial of degree < 2^(extended_k - 1) on the even-indexed extended rows -> on the odd-indexed"""
        self.ctx._ck(self._l.zkhip_extend_half_dev(self.h, C.c_void_p(d_even), C.c_void_p(d_odd), C.c_size_t(batch)))

    def coeff_to_extended_half_dev(self, d_coeff: int, d_out: int, out_stride: int | None = None, batch: int = 1):
        """coeff_to_extended onto the even-indexed extended rows only (the first half of the parity-split layout); members
        land out_stride elements apart (default: back to back)"""
        stride = self.extended_len // 2 if out_stride is None else out_stride
        self.ctx._ck(self._l.zkhip_coeff_to_extended_half_dev(self.h, C.c_void_p(d_coeff), C.c_void_p(d_out),
                                                              C.c_size_t(stride), C.c_size_t(batch)))

    def extended_to_coeff_dev(self, d_ext: int, d_out: int, batch: int = 1):
        self.ctx._ck(self._l.zkhip_extended_to_coeff_dev(self.h, C.c_void_p(d_ext), C.c_void_p(d_out),
                                                         C.c_size_t(batch)))

    def divide_by_vanishing_poly_dev(self, d_ext: int, batch: int = 1):
        self.ctx._ck(self._l.zkhip_divide_by_vanishing_poly_dev(self.h, C.c_void_p(d_ext), C.c_size_t(batch)))

    def _cosets(self, cosets):
        return np.ascontiguousarray(np.asarray(cosets, dtype=np.uint32))

    def coeff_to_cosets_dev(self, d_coeff: int, cosets, d_out: int, batch: int = 1):
        cs = self._cosets(cosets)
        self.ctx._ck(self._l.zkhip_coeff_to_cosets_dev(self.h, C.c_void_p(d_coeff), _p(cs), C.c_uint32(len(cs)),
                                                       C.c_void_p(d_out), C.c_size_t(batch)))

    def extended_to_cosets_dev(self, d_ext: int, cosets, d_out: int):
        cs = self._cosets(cosets)
        self.ctx._ck(self._l.zkhip_extended_to_cosets_dev(self.h, C.c_void_p(d_ext), _p(cs), C.c_uint32(len(cs)), C.c_void_p(d_out)))

    def cosets_to_extended_dev(self, d_compact: int, cosets, d_ext: int):
        cs = self._cosets(cosets)
        self.ctx._ck(self._l.zkhip_cosets_to_extended_dev(self.h, C.c_void_p(d_compact), _p(cs), C.c_uint32(len(cs)), C.c_void_p(d_ext)))

    def permutation_product_dev(self, d_values, d_sigmas, beta, gamma, delta_start, z0, d_z: int):
        """One column set of `permutation::prover::Argument::commit` (all n rows; blinding rows are the caller's)."""
        assert len(d_values) == len(d_sigmas) and d_values
        vals = (C.c_void_p * len(d_values))(*d_values)
        sigs = (C.c_void_p * len(d_sigmas))(*d_sigmas)
        beta, gamma, delta_start, z0 = u64(beta), u64(gamma), u64(delta_start), u64(z0)
        self.ctx._ck(self._l.zkhip_permutation_product_dev(self.h, vals, sigs, C.c_uint32(len(d_values)), _p(beta), _p(gamma),
                                                           _p(delta_start), _p(z0), C.c_void_p(d_z)))

    def permutation_sigma_dev(self, d_map_col: int, d_map_row: int, num_cols: int, d_out: int):
        self.ctx._ck(self._l.zkhip_permutation_sigma_dev(self.h, C.c_void_p(d_map_col), C.c_void_p(d_map_row),
                                                         C.c_uint32(num_cols), C.c_void_p(d_out)))

    def lookup_permute_dev(self, d_input: int, d_table: int, usable_rows: int, d_permuted_input: int, d_permuted_table: int):
        self.ctx._ck(self._l.zkhip_lookup_permute_dev(self.h, C.c_void_p(d_input), C.c_void_p(d_table), C.c_uint32(usable_rows),
                                                      C.c_void_p(d_permuted_input), C.c_void_p(d_permuted_table)))

    def lookup_permute_batch_dev(self, d_inputs, d_tables, usable_rows: int, d_permuted_inputs, d_permuted_tables):
        """lookup_permute_dev for all lookups of a proof in one sequence of launches (lists of device pointers)"""
        count = len(d_inputs)
        assert len(d_tables) == len(d_permuted_inputs) == len(d_permuted_tables) == count
        arr = lambda ptrs: (C.c_void_p * max(count, 1))(*[C.c_void_p(int(p)) for p in ptrs])
        self.ctx._ck(self._l.zkhip_lookup_permute_batch_dev(self.h, C.c_uint32(count), arr(d_inputs), arr(d_tables),
                                                            C.c_uint32(usable_rows), arr(d_permuted_inputs), arr(d_permuted_tables)))

    def lookup_product_dev(self, d_input: int, d_table: int, d_permuted_input: int, d_permuted_table: int, beta, gamma,
                           d_z: int):
        """`lookup::prover::Permuted::commit_product` (all n rows; blinding rows are the caller's)."""
        beta, gamma = u64(beta), u64(gamma)
        self.ctx._ck(self._l.zkhip_lookup_product_dev(self.h, C.c_void_p(d_input), C.c_void_p(d_table),
                                                      C.c_void_p(d_permuted_input), C.c_void_p(d_permuted_table), _p(beta),
                                                      _p(gamma), C.c_void_p(d_z)))

    def lookup_product_batch_dev(self, d_inputs, d_tables, d_permuted_inputs, d_permuted_tables, beta, gamma, d_zs):
        """`lookup_product_dev` for many lookups: the denominators of up to 32 of them are inverted by one launch"""
        beta, gamma = u64(beta), u64(gamma)
        count = len(d_zs)
        assert len(d_inputs) == len(d_tables) == len(d_permuted_inputs) == len(d_permuted_tables) == count
        arr = lambda ps: (C.c_void_p * max(count, 1))(*[C.c_void_p(int(p)) for p in ps])       # noqa: E731
        self.ctx._ck(self._l.zkhip_lookup_product_batch_dev(self.h, arr(d_inputs), arr(d_tables), arr(d_permuted_inputs),
                                                            arr(d_permuted_tables), _p(beta), _p(gamma), arr(d_zs),
                                                            C.c_uint32(count)))

    def evaluate_h_prepare(self, desc: EvalhDesc):
        """`Evaluator::new(cs)`: builds the program of the descriptor's shape and keeps it in the context"""
        self.ctx._ck(self._l.zkhip_evaluate_h_prepare(self.h, C.byref(desc)))

    def evaluate_graphs_dev(self, desc: EvalhDesc, graphs, d_outs):
        """graphs: a ctypes array of Graph over the descriptor's Lagrange-basis device columns; d_outs: one device vector of
        2^k elements per graph.  One call, one wait at its end."""
        outs = (C.c_void_p * max(len(d_outs), 1))(*[C.c_void_p(int(p)) for p in d_outs])
        self.ctx._ck(self._l.zkhip_evaluate_graphs_dev(self.h, C.byref(desc), graphs, C.c_uint32(len(d_outs)), outs))

    def evaluate_h_columns_used(self, desc: EvalhDesc):
        """one flag per column of the program's column table (see zkhip_evaluate_h_columns_used for the order)"""
        cap = (desc.num_fixed + desc.num_advice + desc.num_instance + 4 + desc.num_perm_sets + desc.num_perm_columns
               + 3 * desc.num_lookups)
        out = np.zeros(cap, dtype=np.uint8)
        count = C.c_uint32(0)
        self.ctx._ck(self._l.zkhip_evaluate_h_columns_used(self.h, C.byref(desc), _p(out), C.c_uint32(cap), C.byref(count)))
        return out[:count.value]

    def evaluate_h(self, desc: EvalhDesc, out_dev: int | None = None):
        """`Evaluator::evaluate_h`.  Host descriptor -> numpy result; device descriptor -> writes out_dev."""
        if desc.mem == MEM_HOST:
            rows = self.n if desc.basis else (self.n * desc.num_cosets if desc.num_cosets else self.extended_len)
            rows = desc.row_count or rows
            out = np.empty((rows, 4), dtype=np.uint64)
            self.ctx._ck(self._l.zkhip_evaluate_h(self.h, C.byref(desc), _p(out)))
            return out
        assert out_dev is not None
        self.ctx._ck(self._l.zkhip_evaluate_h(self.h, C.byref(desc), C.c_void_p(out_dev)))
        return None
