"""ctypes binding of oracle/liboracle.so (the C restatement).  TEST INFRASTRUCTURE, not a product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays are numpy uint64 with a trailing axis of 4 limbs (field elements), 8 (affine) or 12 (Jacobian).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_domain_new.restype = C.c_void_p
        _LIB.oracle_domain_extended_k.restype = C.c_uint32
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def field_binop(field, op, a, b):
    a, b = _u64(a), _u64(b)
    out = np.empty_like(a)
    lib().oracle_field_binop(C.c_int(field), C.c_int(op), _p(a), _p(b), _p(out), C.c_size_t(a.size // 4))
    return out


def field_invert(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().oracle_field_invert(C.c_int(field), _p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def to_canonical(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().oracle_field_to_canonical(C.c_int(field), _p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def from_canonical(field, a):
    a = _u64(a)
    out = np.empty_like(a)
    lib().oracle_field_from_canonical(C.c_int(field), _p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def g1_bases(n):
    out = np.empty((n, 8), dtype=np.uint64)
    lib().oracle_g1_bases(_p(out), C.c_size_t(n))
    return out


def g1_powers_of_tau(tau_limbs, n, threads=1):
    """[tau^i] G for i < n (affine); tau_limbs: Montgomery Fr limbs"""
    t = _u64(tau_limbs)
    out = np.empty((n, 8), dtype=np.uint64)
    lib().oracle_g1_powers_of_tau(_p(t), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def g1_to_affine(jac):
    jac = _u64(jac)
    out = np.empty(8, dtype=np.uint64)
    lib().oracle_g1_to_affine(_p(jac), _p(out))
    return out


def g1_add_affine(p, q):
    p, q = _u64(p), _u64(q)
    out = np.empty_like(p)
    lib().oracle_g1_add_affine(_p(p), _p(q), _p(out), C.c_size_t(p.size // 8))
    return out


def g1_sum_jacobian(jacs):
    jacs = _u64(jacs)
    out = np.empty(12, dtype=np.uint64)
    lib().oracle_g1_sum_jacobian(_p(jacs), C.c_size_t(jacs.size // 12), _p(out))
    return out


def best_multiexp(coeffs, bases, threads=1):
    coeffs, bases = _u64(coeffs), _u64(bases)
    n = coeffs.size // 4
    assert bases.size // 8 >= n
    out = np.empty(12, dtype=np.uint64)
    lib().oracle_best_multiexp(_p(coeffs), _p(bases), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def best_fft(a, omega, log_n, threads=1):
    a = _u64(a).copy()
    omega = _u64(omega)
    assert a.size // 4 == 1 << log_n
    lib().oracle_best_fft(_p(a), _p(omega), C.c_uint32(log_n), C.c_int(threads))
    return a


def max_threads():
    return lib().oracle_max_threads()


class Domain:
    """`EvaluationDomain::new(j, k)` on the C oracle."""

    def __init__(self, j, k, threads=1):
        self.j, self.k = j, k
        self.h = C.c_void_p(lib().oracle_domain_new(C.c_uint32(j), C.c_uint32(k), C.c_int(threads)))
        self.extended_k = lib().oracle_domain_extended_k(self.h)
        self.n = 1 << k
        self.extended_len = 1 << self.extended_k
        self.quotient_poly_degree = j - 1

    def __del__(self):
        try:
            lib().oracle_domain_free(self.h)
        except Exception:
            pass

    def constant(self, which):
        out = np.empty(4, dtype=np.uint64)
        lib().oracle_domain_constant(self.h, C.c_int(which), _p(out))
        return out

    def lagrange_to_coeff(self, a):
        a = _u64(a).copy()
        lib().oracle_lagrange_to_coeff(self.h, _p(a))
        return a

    def coeff_to_lagrange(self, a):
        a = _u64(a).copy()
        lib().oracle_coeff_to_lagrange(self.h, _p(a))
        return a

    def coeff_to_extended(self, a):
        a = _u64(a)
        assert a.size // 4 == self.n
        out = np.empty((self.extended_len, 4), dtype=np.uint64)
        lib().oracle_coeff_to_extended(self.h, _p(a), _p(out))
        return out

    def extended_to_coeff(self, a):
        a = _u64(a)
        assert a.size // 4 == self.extended_len
        out = np.empty((self.n * self.quotient_poly_degree, 4), dtype=np.uint64)
        lib().oracle_extended_to_coeff(self.h, _p(a), _p(out))
        return out

    def divide_by_vanishing_poly(self, a):
        a = _u64(a).copy()
        lib().oracle_divide_by_vanishing_poly(self.h, _p(a))
        return a

    def evaluate_h(self, desc_struct):
        """desc_struct: a ctypes zkhip_evalh_desc (host pointers, extended form)."""
        out = np.empty((self.extended_len, 4), dtype=np.uint64)
        rc = lib().oracle_evaluate_h(self.h, C.byref(desc_struct), _p(out))
        if rc != 0:
            raise RuntimeError("oracle_evaluate_h rejected the descriptor")
        return out
