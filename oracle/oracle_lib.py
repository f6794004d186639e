"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes/numpy access to oracle/c/liboracle.so (the plain-C CPU restatement of the reference's
third-party arithmetic: bellman_ce 0.3.2 @ 5809cc16 / pairing_ce 0.24.2 / ff_ce 0.12.0, see
oracle/c/oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (plonkit_amd/) never does.

Data conventions (same as the product's C ABI):
  Fr / Fq vector  = numpy uint64 array of shape [n, 4], little-endian limbs, Montgomery form
  G1 affine vector= uint64 [n, 8]  (x || y), infinity = all zero
  G1 jacobian     = uint64 [n, 12] (X || Y || Z), infinity = Z == 0
Python ints are always canonical (non-Montgomery).
"""
import ctypes
import os
import subprocess

import numpy as np

R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001  # Fr
Q_MOD = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47  # Fq
MONT_R = 1 << 256
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle.so")


def build(force=False):
    """Compile oracle/c with gcc (recipe: oracle/c/Makefile)."""
    src = [os.path.join(_HERE, "c", f) for f in ("oracle.c", "bn254.h", "Makefile")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None
C_SECONDS = [0.0]          # wall-clock spent inside liboracle.so calls (bench.py: the C share of a CPU prove)


class _Timed:
    """the CDLL with every call timed: bench.py's cpu_baseline reports how much of the oracle's prove is the C
    arithmetic (MSM, NTT, vector ops) and how much is the numpy / Python glue around it"""

    def __init__(self, cdll):
        self._cdll = cdll
        self._cache = {}

    def __getattr__(self, name):
        f = self._cache.get(name)
        if f is None:
            raw = getattr(self._cdll, name)
            import time as _t

            def f(*a, _raw=raw):
                t0 = _t.perf_counter()
                try:
                    return _raw(*a)
                finally:
                    C_SECONDS[0] += _t.perf_counter() - t0
            f.raw = raw
            self._cache[name] = f
        return f


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _Timed(ctypes.CDLL(_SO))
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_THREADS = [None]


def set_threads(n):
    """thread count of every call that does not name one (None: back to the default); bench.py's cpu_baseline leg sets it"""
    _THREADS[0] = int(n) if n else None


def ncpu():
    """default thread count of the checker (tests: at most 16); the cpu_baseline leg sets its own with set_threads()"""
    return _THREADS[0] or min(os.cpu_count() or 1, 16)


# ----------------------------------------------------------------- int <-> limb conversions
def int_to_limbs(x):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a):
    return sum(int(a[i]) << (64 * i) for i in range(4))


def ints_to_array(xs):
    """list of canonical ints -> uint64 [n,4] (still canonical)."""
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for i, x in enumerate(xs):
        for j in range(4):
            out[i, j] = (x >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def array_to_ints(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def fr_mont(x):
    """canonical int -> Montgomery limbs (pure Python, independent of the C code)."""
    return int_to_limbs((x % R_MOD) * MONT_R % R_MOD)


def fq_mont(x):
    return int_to_limbs((x % Q_MOD) * MONT_R % Q_MOD)


def fr_vec(xs):
    """list of canonical ints -> Montgomery Fr array [n,4]."""
    return ints_to_array([(x % R_MOD) * MONT_R % R_MOD for x in xs])


def fr_ints(a):
    """Montgomery Fr array -> list of canonical ints."""
    rinv = pow(MONT_R, -1, R_MOD)
    return [v * rinv % R_MOD for v in array_to_ints(a)]


def fq_ints(a):
    rinv = pow(MONT_R, -1, Q_MOD)
    return [v * rinv % Q_MOD for v in array_to_ints(a)]


def fr_zeros(n):
    return np.zeros((n, 4), dtype=np.uint64)


# ----------------------------------------------------------------------------- Fr vectors
def omega(log_n):
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_omega(_p(out), ctypes.c_uint32(log_n))
    return fr_ints(out)[0]


def ntt(a, log_n, inverse=False, coset=None, threads=None):
    """in-place on a copy; natural order in/out; coset = canonical int or None."""
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    assert a.shape == (1 << log_n, 4)
    c = fr_mont(coset) if coset is not None else None
    lib().orc_fr_ntt(_p(a), ctypes.c_uint32(log_n), ctypes.c_int(1 if inverse else 0),
                     _p(c) if c is not None else None, ctypes.c_int(threads or ncpu()))
    return a


def dft_naive(a, log_n):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    lib().orc_fr_dft_naive(_p(out), _p(a), ctypes.c_uint32(log_n))
    return out


def _vec2(name, a, b, threads=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    assert a.shape == b.shape
    out = np.empty_like(a)
    getattr(lib(), name)(_p(out), _p(a), _p(b), ctypes.c_uint64(a.shape[0]), ctypes.c_int(threads or ncpu()))
    return out


def vmul(a, b):
    return _vec2("orc_fr_vec_mul", a, b)


def vadd(a, b):
    return _vec2("orc_fr_vec_add", a, b)


def vsub(a, b):
    return _vec2("orc_fr_vec_sub", a, b)


def vscale(a, s):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    sm = fr_mont(s)
    lib().orc_fr_vec_scale(_p(out), _p(a), _p(sm), ctypes.c_uint64(a.shape[0]), ctypes.c_int(ncpu()))
    return out


def vadd_scalar(a, s):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    sm = fr_mont(s)
    lib().orc_fr_vec_add_scalar(_p(out), _p(a), _p(sm), ctypes.c_uint64(a.shape[0]), ctypes.c_int(ncpu()))
    return out


def vaxpy(a, s, b):
    """a + s*b"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    sm = fr_mont(s)
    lib().orc_fr_vec_axpy(_p(out), _p(a), _p(sm), _p(b), ctypes.c_uint64(a.shape[0]), ctypes.c_int(ncpu()))
    return out


def vpowers(g, n, s=1):
    """[s, s*g, s*g^2, ...]"""
    out = fr_zeros(n)
    lib().orc_fr_vec_powers(_p(out), _p(fr_mont(g)), _p(fr_mont(s)), ctypes.c_uint64(n))
    return out


def vbatch_inv(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().orc_fr_vec_batch_inv(_p(a), ctypes.c_uint64(a.shape[0]))
    return a


def vshifted_prefix_product(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.zeros_like(a)
    lib().orc_fr_vec_shifted_prefix_product(_p(out), _p(a), ctypes.c_uint64(a.shape[0]))
    return out


def poly_eval(c, x):
    c = np.ascontiguousarray(c, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_fr_poly_eval(_p(out), _p(c), ctypes.c_uint64(c.shape[0]), _p(fr_mont(x)))
    return fr_ints(out)[0]


def poly_div_linear(p, z):
    """(p(x) - p(z)) / (x - z); same length, top coefficient 0."""
    p = np.ascontiguousarray(p, dtype=np.uint64)
    q = np.zeros_like(p)
    lib().orc_fr_poly_div_linear(_p(q), _p(p), ctypes.c_uint64(p.shape[0]), _p(fr_mont(z)))
    return q


# --------------------------------------------------------------------------------- G1
def g1_generator():
    out = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g1_from_ints(x, y):
    """canonical affine coordinates -> Montgomery affine limbs; (0,0) = infinity."""
    return np.concatenate([fq_mont(x), fq_mont(y)])


def g1_to_ints(p):
    p = np.ascontiguousarray(p, dtype=np.uint64).reshape(8)
    x, y = fq_ints(p.reshape(2, 4))
    return x, y


def g1_is_inf(p):
    return not np.any(np.asarray(p))


def g1_add(a, b):
    out = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_add_affine(_p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))
    return out


def g1_neg(a):
    out = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_neg_affine(_p(out), _p(np.ascontiguousarray(a)))
    return out


def g1_mul(a, k):
    out = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_mul_affine(_p(out), _p(np.ascontiguousarray(a)), _p(fr_mont(k)))
    return out


def g1_on_curve(a):
    lib().orc_g1_on_curve.raw.restype = ctypes.c_int
    return bool(lib().orc_g1_on_curve(_p(np.ascontiguousarray(a))))


def jac_to_affine(j):
    j = np.ascontiguousarray(j, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((j.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_jac_to_affine(_p(out), _p(j), ctypes.c_uint64(j.shape[0]))
    return out


def crs42(n, threads=None):
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().orc_crs42(_p(out), ctypes.c_uint64(n), ctypes.c_int(threads or ncpu()))
    return out


MSM_SPLIT = ["chunks"]       # default work split of msm(): "chunks" = bellman 0.3.2's dense_multiexp, "windows" = (window, chunk) tasks


def msm(bases, scalars, threads=None, naive=False, split=None):
    """sum scalars[i]*bases[i] -> affine limbs [8].  bellman dense_multiexp restatement; split="windows": the same sum with
    the work cut by (window, chunk) as later bellman revisions do (orc_g1_msm_wc: what scales past ~32 threads)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.shape[0]
    assert bases.shape[0] >= n
    out = np.zeros(12, dtype=np.uint64)
    if naive:
        lib().orc_g1_msm_naive(_p(out), _p(bases), _p(scalars), ctypes.c_uint64(n))
    elif (split or MSM_SPLIT[0]) == "windows":
        lib().orc_g1_msm_wc(_p(out), _p(bases), _p(scalars), ctypes.c_uint64(n), ctypes.c_int(threads or ncpu()))
    else:
        lib().orc_g1_msm(_p(out), _p(bases), _p(scalars), ctypes.c_uint64(n), ctypes.c_int(threads or ncpu()))
    return jac_to_affine(out)[0]


def msm_jacobian(bases, scalars, threads=None):
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    lib().orc_g1_msm(_p(out), _p(bases), _p(scalars), ctypes.c_uint64(scalars.shape[0]),
                     ctypes.c_int(threads or ncpu()))
    return out


def g1_intt(points, log_n, threads=None):
    points = np.ascontiguousarray(points, dtype=np.uint64)
    out = np.zeros((1 << log_n, 8), dtype=np.uint64)
    lib().orc_g1_intt(_p(out), _p(points), ctypes.c_uint32(log_n), ctypes.c_int(threads or ncpu()))
    return out


# ------------------------------------------------------- circuit front end in C (CPU baseline leg)
def r1cs_parse(data: bytes):
    """iden3 .r1cs -> (header dict, off uint64[3 nc + 1], wires uint32[nt], coeffs Montgomery [nt, 4])"""
    hdr = np.zeros(5, dtype=np.uint64)
    L = lib()
    L.orc_r1cs_parse.raw.restype = ctypes.c_int
    rc = L.orc_r1cs_parse(data, ctypes.c_uint64(len(data)), _p(hdr), None, None, None)
    if rc != 0:
        raise ValueError("InvalidData: malformed r1cs (%d)" % rc)
    n_wires, n_pub_out, n_pub_in, nc, nt = (int(x) for x in hdr)
    off = np.zeros(3 * nc + 1, dtype=np.uint64)
    wires = np.zeros(max(nt, 1), dtype=np.uint32)
    coeffs = np.zeros((max(nt, 1), 4), dtype=np.uint64)
    rc = L.orc_r1cs_parse(data, ctypes.c_uint64(len(data)), _p(hdr), _p(off), _p(wires), _p(coeffs))
    if rc != 0:
        raise ValueError("InvalidData: malformed r1cs (%d)" % rc)
    return dict(n_wires=n_wires, n_pub_out=n_pub_out, n_pub_in=n_pub_in, n_constraints=nc), off, wires, coeffs


def wtns_parse(data: bytes):
    """iden3 .wtns -> Montgomery Fr array [n, 4]"""
    L = lib()
    L.orc_wtns_parse.raw.restype = ctypes.c_int64
    n = L.orc_wtns_parse(data, ctypes.c_uint64(len(data)), None, ctypes.c_uint64(0))
    if n < 0:
        raise ValueError("malformed wtns (%d)" % n)
    out = np.zeros((n, 4), dtype=np.uint64)
    if L.orc_wtns_parse(data, ctypes.c_uint64(len(data)), _p(out), ctypes.c_uint64(n)) != n:
        raise ValueError("malformed wtns")
    return out


def transpile_c(off, wires, coeffs, num_variables, witness=None, first_row=0, want_q=True, cap=None):
    """gate rows (vars uint32 [4, cap], q Montgomery [7, cap, 4] or None), number of gates, values [num_vars, 4] or None"""
    nc = (off.shape[0] - 1) // 3
    nt = int(off[-1])
    cap = cap or (first_row + nc + nt + 8)                     # a constraint makes at most (terms + 3) gates
    cap_vals = num_variables + 2 * nt + 3 * nc + 8
    vars_ = np.zeros((4, cap), dtype=np.uint32)
    q = np.zeros((7, cap, 4), dtype=np.uint64) if want_q else None
    values = np.zeros((cap_vals, 4), dtype=np.uint64) if witness is not None else None
    nv = ctypes.c_uint64(0)
    L = lib()
    L.orc_transpile.raw.restype = ctypes.c_int64
    if witness is not None:
        witness = np.ascontiguousarray(witness, dtype=np.uint64)
        assert witness.shape[0] >= num_variables, "witness shorter than the number of variables"
    g = L.orc_transpile(_p(off), _p(wires), _p(coeffs), ctypes.c_uint64(nc), ctypes.c_uint64(num_variables),
                        _p(witness) if witness is not None else None, _p(vars_), _p(q) if want_q else None, ctypes.c_uint64(cap),
                        ctypes.c_uint64(first_row), _p(values) if values is not None else None, ctypes.c_uint64(cap_vals), ctypes.byref(nv))
    if g == -1:
        raise AssertionError("unsatisfiable constant constraint")
    if g < 0:
        raise MemoryError("orc_transpile: capacity exceeded")
    return vars_, q, int(g), (values[:nv.value] if values is not None else None), int(nv.value)


def gather_columns(vars_, values, n):
    """cols [4, n, 4]: cols[j][r] = values[vars[j][r]]"""
    cols = np.empty((4, n, 4), dtype=np.uint64)
    lib().orc_gather_columns(_p(cols), _p(vars_), ctypes.c_uint64(vars_.shape[1]), _p(values), ctypes.c_uint64(n), ctypes.c_int(ncpu()))
    return cols


def check_gates(cols, sel, n, num_inputs):
    L = lib()
    L.orc_check_gates.raw.restype = ctypes.c_int
    cols = np.ascontiguousarray(cols, dtype=np.uint64); sel = np.ascontiguousarray(sel, dtype=np.uint64)
    return bool(L.orc_check_gates(_p(cols), _p(sel), ctypes.c_uint64(n), ctypes.c_uint64(num_inputs), ctypes.c_int(ncpu())))


def keccak256(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().orc_keccak256(out, data, ctypes.c_uint64(len(data)))
    return out.raw
