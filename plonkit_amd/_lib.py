"""ctypes binding of include/plonkit_amd.h (lib/libplonkit_amd.so).

Arrays cross as numpy uint64 ([n,4] Fr Montgomery, [n,8] G1 affine, [12] Jacobian) or as raw device
pointers (ints / torch tensors).  Nothing here computes: every function forwards to the C ABI, and
a missing library or a missing GPU raises — there is no Python or CPU fallback.
"""
import ctypes
import sys
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib", "libplonkit_amd.so")
_lib = None

ERR_NAMES = {1: "PLK_ERR_ARG", 2: "PLK_ERR_SIZE", 3: "PLK_ERR_SRS", 4: "PLK_ERR_HIP", 5: "PLK_ERR_UNSAT",
             6: "PLK_ERR_FORMAT", 7: "PLK_ERR_IO"}


class PlkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


def lib_path():
    return _SO


def lib():
    """Loads the shared library; raises loudly if it has not been built (python -m plonkit_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("plonkit_amd: %s is missing — build it with `python -m plonkit_amd.build` "
                              "(hipcc, gfx950). There is no fallback implementation." % _SO)
        # PyTorch-ROCm ships its own libamdhip64; if /opt/rocm's copy (our DT_NEEDED) is mapped first, torch's
        # later HIP initialisation finds "No HIP GPUs".  Import torch first so that one runtime serves both.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(_SO)
        L.plk_last_error.restype = ctypes.c_char_p
        L.plk_version.restype = ctypes.c_char_p
        L.plk_srs_size.restype = ctypes.c_uint64
        if hasattr(L, "plk_setup_domain_size"):
            L.plk_setup_domain_size.restype = ctypes.c_uint64
        _lib = L
    return _lib


def last_error():
    return lib().plk_last_error().decode()


def _check(rc):
    if rc != 0:
        raise PlkError(rc, last_error())


def have_gpu():
    return lib().plk_device_count() > 0


def _np(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _devptr(x):
    """torch tensor / int -> void*"""
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(int(x))


def _stream(s):
    if s is None:
        return ctypes.c_void_p(0)
    if hasattr(s, "cuda_stream"):
        return ctypes.c_void_p(s.cuda_stream)
    return ctypes.c_void_p(int(s))


class Context:
    """plk_ctx: one GPU.  Mirrors where the reference builds `Worker::new()` (src/plonk.rs:41,47,183)."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        _check(lib().plk_create(ctypes.c_int32(device), ctypes.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            lib().plk_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().plk_synchronize(self._h))

    # ---- SRS
    def srs_upload(self, bases):
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        assert bases.ndim == 2 and bases.shape[1] == 8
        _check(lib().plk_srs_upload(self._h, _np(bases), ctypes.c_uint64(bases.shape[0])))

    def srs_set_dev(self, ptr, n):
        _check(lib().plk_srs_set_dev(self._h, _devptr(ptr), ctypes.c_uint64(n)))

    def srs_size(self):
        return lib().plk_srs_size(self._h)

    # multi-GPU commitments of the prover: this rank's SRS slice starts at global index `first_index`; `combine`
    # receives a writable uint64[count, 12] array of Jacobian partial sums and must replace it by the all-ranks sums
    def set_commit_shard(self, first_index, combine):
        if combine is None:
            self._combine_cb = None
            _check(lib().plk_set_commit_shard(self._h, ctypes.c_uint64(0), None, None))
            return
        proto = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32)

        def trampoline(_user, ptr, count):
            try:
                combine(np.ctypeslib.as_array(ptr, shape=(count, 12)))
                return 0
            except Exception as exc:                                  # never let an exception cross the C boundary
                sys.stderr.write("commit combiner failed: %r\n" % (exc,))
                return 4
        self._combine_cb = proto(trampoline)                          # keep the callback object alive
        _check(lib().plk_set_commit_shard(self._h, ctypes.c_uint64(first_index), self._combine_cb, None))

    # the same exchange built into the library (comm.cpp): RCCL all-gather + host EC sum, no Python in the loop
    def comm_init(self, rank, world, unique_id, first_index):
        assert len(unique_id) == 128
        _check(lib().plk_comm_init(self._h, ctypes.c_int32(rank), ctypes.c_int32(world), bytes(unique_id), ctypes.c_uint64(first_index)))

    def comm_init_tcp(self, rank, world, port, first_index):
        _check(lib().plk_comm_init_tcp(self._h, ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.c_uint16(port), ctypes.c_uint64(first_index)))

    def comm_set_mode(self, mode):
        """"replicate" (every rank proves, commitments split) or "scatter" (owner computes: rank 0 proves, the others comm_serve); every
        rank of the communicator must choose the same"""
        _check(lib().plk_comm_set_mode(self._h, ctypes.c_int32({"replicate": 0, "scatter": 1}[mode])))

    def comm_serve(self):
        """worker ranks of a scatter-mode communicator: commit what the owner sends until it calls comm_stop_workers; returns the batches served"""
        n = ctypes.c_uint64(0)
        _check(lib().plk_comm_serve(self._h, ctypes.byref(n)))
        return int(n.value)

    def comm_stop_workers(self):
        _check(lib().plk_comm_stop_workers(self._h))

    def comm_set_shard(self, first_index):
        _check(lib().plk_comm_set_shard(self._h, ctypes.c_uint64(first_index)))

    def msm_finish_sharded(self):
        """finish + the built-in exchange: affine commitment over all ranks' shards"""
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().plk_msm_g1_finish_sharded(self._h, _np(out)))
        return out

    def comm_selftest(self):
        """header broadcast + one grouped ring step over the RCCL communicator, checked byte by byte (every rank calls it)"""
        _check(lib().plk_comm_selftest(self._h))

    def comm_scatter_selftest(self, log_n, iterations):
        """the scatter step of owner-computes mode through the real RCCL branch on one GPU (one-rank communicator, rank 0 receiving its own
        share), the scalars still being written when the step is called; returns the number of iterations whose commitment differs (0 = pass)"""
        bad = ctypes.c_uint32(0)
        _check(lib().plk_comm_scatter_selftest(self._h, ctypes.c_uint32(log_n), ctypes.c_uint32(iterations), ctypes.byref(bad)))
        return bad.value

    def comm_nccl_count(self):
        """ranks RCCL itself counts in this context's communicator (ncclCommCount); 0 without an RCCL communicator"""
        k = ctypes.c_int32(0)
        _check(lib().plk_comm_nccl_count(self._h, ctypes.byref(k)))
        return k.value

    def comm_destroy(self):
        _check(lib().plk_comm_destroy(self._h))

    def comm_info(self):
        r, w, x = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_uint64(0)
        _check(lib().plk_comm_info(self._h, ctypes.byref(r), ctypes.byref(w), ctypes.byref(x)))
        return r.value, w.value, x.value

    # Lagrange-form key (`prove -l`): second resident SRS used by prove() for commit_using_values
    def srs_lagrange_upload(self, bases):
        bases = np.ascontiguousarray(bases, dtype=np.uint64)
        assert bases.ndim == 2 and bases.shape[1] == 8
        _check(lib().plk_srs_lagrange_upload(self._h, _np(bases), ctypes.c_uint64(bases.shape[0])))

    def srs_lagrange_set_dev(self, ptr, n):
        _check(lib().plk_srs_lagrange_set_dev(self._h, _devptr(ptr), ctypes.c_uint64(n)))

    def srs_lagrange_clear(self):
        _check(lib().plk_srs_lagrange_clear(self._h))

    def srs_lagrange_size(self):
        return lib().plk_srs_lagrange_size(self._h)

    def srs_generate(self, n, start=0, tau=42):
        """Crs::crs_42 on the GPU (src/plonk.rs:30-48): resident SRS <- tau^(start+i) * G."""
        _check(lib().plk_srs_generate(self._h, ctypes.c_uint64(n), ctypes.c_uint64(start), ctypes.c_uint32(tau)))

    def srs_generate_fr(self, n, start, tau_mont):
        t = np.ascontiguousarray(tau_mont, dtype=np.uint64)
        _check(lib().plk_srs_generate_fr(self._h, ctypes.c_uint64(n), ctypes.c_uint64(start), _np(t)))

    def srs_precompute(self):
        """builds the MSM's fixed-base table of the resident key(s) now instead of at the first commitment"""
        _check(lib().plk_srs_precompute(self._h))

    def share_srs_from(self, owner):
        """borrow `owner`'s resident key(s) and MSM fixed-base table(s) (same device): a second context for proofs in flight
        beside `owner`'s costs workspace only.  `owner` must outlive this context's use of the key."""
        _check(lib().plk_ctx_share_srs(self._h, owner._h))
        self._srs_owner = owner                                       # keep the lender alive

    def srs_download(self, offset, n):
        out = np.zeros((n, 8), dtype=np.uint64)
        _check(lib().plk_srs_download(self._h, ctypes.c_uint64(offset), ctypes.c_uint64(n), _np(out)))
        return out

    def set_kernel_timing(self, on=True):
        _check(lib().plk_set_kernel_timing(self._h, ctypes.c_int32(1 if on else 0)))

    def msm_last_kernel_ms(self):
        v = ctypes.c_float(0)
        _check(lib().plk_msm_last_kernel_ms(self._h, ctypes.byref(v)))
        return v.value

    # ---- NTT
    def ntt(self, data, log_n, inverse=False, coset=None):
        """host array in, new host array out (natural order)."""
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        assert a.shape == (1 << log_n, 4)
        c = np.ascontiguousarray(coset, dtype=np.uint64) if coset is not None else None
        _check(lib().plk_ntt(self._h, _np(a), ctypes.c_uint32(log_n), ctypes.c_int32(1 if inverse else 0),
                             _np(c) if c is not None else None))
        return a

    def ntt_dev(self, ptr, log_n, inverse=False, coset=None, stream=None):
        c = np.ascontiguousarray(coset, dtype=np.uint64) if coset is not None else None
        _check(lib().plk_ntt_dev(self._h, _devptr(ptr), ctypes.c_uint32(log_n), ctypes.c_int32(1 if inverse else 0),
                                 _np(c) if c is not None else None, _stream(stream)))

    def lde4(self, coeffs, log_n):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        assert a.shape == (1 << log_n, 4)
        out = np.zeros((4 << log_n, 4), dtype=np.uint64)
        _check(lib().plk_lde4(self._h, _np(a), ctypes.c_uint32(log_n), _np(out)))
        return out

    def lde4_dev(self, in_ptr, log_n, out_ptr, stream=None):
        _check(lib().plk_lde4_dev(self._h, _devptr(in_ptr), ctypes.c_uint32(log_n), _devptr(out_ptr), _stream(stream)))

    def lde4_coset_major_dev(self, in_ptrs, log_n, out_ptrs, stream=None):
        """`count` polynomials -> their 4n evaluations in the prover's coset-major order (out[k*n + r] = f(7 w_4n^(4r+k)))"""
        cnt = len(in_ptrs)
        ins = (ctypes.c_void_p * cnt)(*[_devptr(p).value for p in in_ptrs])
        outs = (ctypes.c_void_p * cnt)(*[_devptr(p).value for p in out_ptrs])
        _check(lib().plk_lde4_coset_major_dev(self._h, ins, ctypes.c_uint32(cnt), ctypes.c_uint32(log_n), outs, _stream(stream)))

    def icoset4_coset_major_dev(self, ptr, log_n, stream=None):
        """4n values in coset-major order -> the 4n coefficients (natural order), in place"""
        _check(lib().plk_icoset4_coset_major_dev(self._h, _devptr(ptr), ctypes.c_uint32(log_n), _stream(stream)))

    # ---- MSM
    def msm(self, scalars, base_offset=0):
        s = np.ascontiguousarray(scalars, dtype=np.uint64)
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().plk_msm_g1(self._h, _np(s), ctypes.c_uint64(s.shape[0]), ctypes.c_uint64(base_offset), _np(out)))
        return out

    def msm_dev(self, ptr, n, base_offset=0, stream=None):
        out = np.zeros(8, dtype=np.uint64)
        _check(lib().plk_msm_g1_dev(self._h, _devptr(ptr), ctypes.c_uint64(n), ctypes.c_uint64(base_offset), _np(out), _stream(stream)))
        return out

    def msm_batch_dev(self, ptrs, n, base_offset=0, stream=None):
        """several commitments (same length, same bases) in one pass; returns [count, 8] affine points"""
        arr = (ctypes.c_void_p * len(ptrs))(*[_devptr(p) for p in ptrs])
        out = np.zeros((len(ptrs), 8), dtype=np.uint64)
        _check(lib().plk_msm_g1_batch_dev(self._h, arr, ctypes.c_uint32(len(ptrs)), ctypes.c_uint64(n), ctypes.c_uint64(base_offset), _np(out), _stream(stream)))
        return out

    def msm_partial_dev(self, ptr, n, base_offset=0, stream=None):
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().plk_msm_g1_partial_dev(self._h, _devptr(ptr), ctypes.c_uint64(n), ctypes.c_uint64(base_offset), _np(out), _stream(stream)))
        return out

    def msm_enqueue_dev(self, ptr, n, base_offset=0, stream=None):
        _check(lib().plk_msm_g1_enqueue_dev(self._h, _devptr(ptr), ctypes.c_uint64(n), ctypes.c_uint64(base_offset), _stream(stream)))

    def msm_enqueue_batch_dev(self, ptrs, n, base_offset=0, stream=None):
        arr = (ctypes.c_void_p * len(ptrs))(*[_devptr(p) for p in ptrs])
        _check(lib().plk_msm_g1_enqueue_batch_dev(self._h, arr, ctypes.c_uint32(len(ptrs)), ctypes.c_uint64(n), ctypes.c_uint64(base_offset), _stream(stream)))

    def msm_finish_batch(self, count):
        out = np.zeros((count, 12), dtype=np.uint64)
        _check(lib().plk_msm_g1_finish_batch(self._h, _np(out), ctypes.c_uint32(count)))
        return out

    def msm_finish_batch_sharded(self, count):
        """finish + the installed combiner (one exchange for the batch): [count, 8] affine commitments over all ranks' shards"""
        out = np.zeros((count, 8), dtype=np.uint64)
        _check(lib().plk_msm_g1_finish_batch_sharded(self._h, _np(out), ctypes.c_uint32(count)))
        return out

    def msm_finish(self):
        out = np.zeros(12, dtype=np.uint64)
        _check(lib().plk_msm_g1_finish(self._h, _np(out)))
        return out

    def g1_intt(self, points, log_n):
        p = np.ascontiguousarray(points, dtype=np.uint64)
        assert p.shape == (1 << log_n, 8)
        out = np.zeros_like(p)
        _check(lib().plk_g1_intt(self._h, _np(p), ctypes.c_uint32(log_n), _np(out)))
        return out


    # ---- tracing hook and the polynomial helpers of rounds 2, 4, 5 (tests localise a wrong proof with them)
    def prove_trace(self, which):
        """vector `which` of the last prove on this context: 0..3 wire polynomials, 4 z, 5 t (4N), 6 r, 7 / 8 opening quotients"""
        n = ctypes.c_uint64(0)
        _check(lib().plk_prove_trace(self._h, ctypes.c_uint32(which), None, ctypes.c_uint64(0), ctypes.byref(n)))
        out = np.zeros((n.value, 4), dtype=np.uint64)
        _check(lib().plk_prove_trace(self._h, ctypes.c_uint32(which), _np(out), ctypes.c_uint64(n.value), ctypes.byref(n)))
        return out

    def poly_evaluate_at_dev(self, ptr, n, z, stream=None):
        z = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        _check(lib().plk_poly_evaluate_at_dev(self._h, _devptr(ptr), ctypes.c_uint64(n), _np(z), _np(out), _stream(stream)))
        return out

    def poly_divide_by_linear_dev(self, ptr, n, z, out_ptr, stream=None):
        z = np.ascontiguousarray(z, dtype=np.uint64)
        _check(lib().plk_poly_divide_by_linear_dev(self._h, _devptr(ptr), ctypes.c_uint64(n), _np(z), _devptr(out_ptr), _stream(stream)))

    def permutation_grand_product_dev(self, wires, sigmas, beta, gamma, log_n, out_ptr, stream=None):
        w = (ctypes.c_void_p * 4)(*[_devptr(p) for p in wires])
        s = (ctypes.c_void_p * 4)(*[_devptr(p) for p in sigmas])
        b, g = np.ascontiguousarray(beta, dtype=np.uint64), np.ascontiguousarray(gamma, dtype=np.uint64)
        _check(lib().plk_permutation_grand_product_dev(self._h, w, s, _np(b), _np(g), ctypes.c_uint32(log_n), _devptr(out_ptr), _stream(stream)))

    def g1_intt_srs_dev(self, log_n, out_ptr, stream=None):
        _check(lib().plk_g1_intt_srs_dev(self._h, ctypes.c_uint32(log_n), _devptr(out_ptr), _stream(stream)))


# ------------------------------------------------- circuit pipeline (mirrors src/plonk.rs's API)
class Circuit:
    """CircomCircuit{r1cs, witness, wire_mapping: None, aux_offset: 1} (src/circom_circuit.rs:41-47).
    File type is chosen from the suffix exactly as the reference does (src/reader.rs:93,179)."""

    def __init__(self, r1cs_bytes, r1cs_is_json, witness_bytes=None, witness_is_json=False):
        self._h = ctypes.c_void_p()
        _check(lib().plk_circuit_load(bytes(r1cs_bytes), ctypes.c_uint64(len(r1cs_bytes)), ctypes.c_int32(1 if r1cs_is_json else 0),
                                      bytes(witness_bytes) if witness_bytes is not None else None,
                                      ctypes.c_uint64(len(witness_bytes) if witness_bytes is not None else 0),
                                      ctypes.c_int32(1 if witness_is_json else 0), ctypes.byref(self._h)))

    @classmethod
    def synthetic(cls, target_gates, seed=0x706c6f6e6b6974):
        """seeded chain circuit with exactly `target_gates` gates and one public input (bench input)"""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        _check(lib().plk_circuit_synthetic(ctypes.c_uint64(target_gates), ctypes.c_uint64(seed), ctypes.byref(self._h)))
        return self

    @classmethod
    def synthetic_ex(cls, target_gates, seed=0x706c6f6e6b6974, witness_seed=0, lc_terms=0):
        """the same generator; witness_seed != 0: same R1CS, another satisfying witness; lc_terms >= 5: dense body (long
        linear combinations folded through the d column: all 11 commitments of a proof are non-trivial; parity unpinned)"""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        _check(lib().plk_circuit_synthetic_ex(ctypes.c_uint64(target_gates), ctypes.c_uint64(seed), ctypes.c_uint64(witness_seed),
                                              ctypes.c_uint32(lc_terms), ctypes.byref(self._h)))
        return self

    def export(self, what):
        """what = "r1cs" | "wtns": bytes in the reference's binary formats"""
        code = {"r1cs": 0, "wtns": 1}[what]
        n = ctypes.c_uint64(0)
        _check(lib().plk_circuit_export(self._h, ctypes.c_int32(code), None, ctypes.c_uint64(0), ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        _check(lib().plk_circuit_export(self._h, ctypes.c_int32(code), buf, ctypes.c_uint64(n.value), ctypes.byref(n)))
        return buf.raw

    @classmethod
    def from_files(cls, r1cs_path, witness_path=None):
        w = open(witness_path, "rb").read() if witness_path else None
        return cls(open(r1cs_path, "rb").read(), r1cs_path.endswith("json"), w, bool(witness_path) and witness_path.endswith("json"))

    def domain_size(self):
        """N of the circuit's setup without building it (transpile only)"""
        n = ctypes.c_uint64(0)
        _check(lib().plk_circuit_domain_size(self._h, ctypes.byref(n)))
        return n.value

    def analyse(self):
        """plonk::analyse (src/plonk.rs:72-93) as the serde_json string of src/tests.rs:14"""
        size = 1 << 20
        while True:                                                   # ~40 bytes per constraint: grow until it fits
            buf = ctypes.create_string_buffer(size)
            rc = lib().plk_circuit_analyse(self._h, buf, ctypes.c_uint64(size))
            if rc == 0:
                return buf.value.decode()
            if size >= (1 << 31) or "too small" not in last_error():
                _check(rc)
            size <<= 3

    def close(self):
        if self._h:
            lib().plk_circuit_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SetupForProver:
    """SetupForProver (src/plonk.rs:50-176): prepare_setup_for_prover / make_verification_key / prove."""

    def __init__(self, ctx, circuit):
        self.ctx = ctx
        self._h = ctypes.c_void_p()
        _check(lib().plk_setup_prepare(ctx._h, circuit._h, ctypes.byref(self._h)))

    @classmethod
    def prepare_host(cls, circuit):
        """the CPU half only (transpile + columns): no context, no GPU; finish with upload(ctx)"""
        self = cls.__new__(cls)
        self.ctx = None
        self._h = ctypes.c_void_p()
        _check(lib().plk_setup_prepare_host(circuit._h, ctypes.byref(self._h)))
        return self

    def upload(self, ctx):
        _check(lib().plk_setup_upload(ctx._h, self._h))
        self.ctx = ctx
        return self

    @property
    def domain_size(self):
        return lib().plk_setup_domain_size(self._h)

    def verification_key_bytes(self, g2_bytes):
        assert len(g2_bytes) == 256
        out = ctypes.create_string_buffer(4096)
        n = ctypes.c_uint64(0)
        _check(lib().plk_setup_write_vk(self.ctx._h, self._h, bytes(g2_bytes), out, ctypes.c_uint64(len(out)), ctypes.byref(n)))
        return out.raw[:n.value]

    def prove(self, circuit, ctx=None):
        """proof.bin bytes (keccak transcript, monomial key — src/plonk.rs:152-159).  `ctx`: another context on the same device
        (one per host thread: several proofs of this setup may be in flight at once; ctypes releases the GIL for the call)"""
        h = (ctx or self.ctx)._h
        cap = 1 << 16
        out = ctypes.create_string_buffer(cap)
        n = ctypes.c_uint64(0)
        rc = lib().plk_prove(h, self._h, circuit._h, out, ctypes.c_uint64(cap), ctypes.byref(n))
        if rc == 1 and n.value > cap:                                 # many public inputs: retry with the reported size
            cap = n.value
            out = ctypes.create_string_buffer(cap)
            rc = lib().plk_prove(h, self._h, circuit._h, out, ctypes.c_uint64(cap), ctypes.byref(n))
        _check(rc)
        return out.raw[:n.value]

    def timings_ms(self, ctx=None):
        arr = (ctypes.c_double * 16)()
        cnt = ctypes.c_uint32(0)
        _check(lib().plk_prove_timings((ctx or self.ctx)._h, arr, ctypes.c_uint32(16), ctypes.byref(cnt)))
        names = ["witness", "round1", "round2", "round3", "round4", "round5", "serialise"]
        return {names[i] if i < len(names) else str(i): arr[i] for i in range(cnt.value)}

    def close(self):
        if self._h:
            lib().plk_setup_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ CPU-only helpers of the ABI
def g1_sum_jacobian(parts):
    parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(8, dtype=np.uint64)
    _check(lib().plk_g1_sum_jacobian(_np(parts), ctypes.c_uint64(parts.shape[0]), _np(out)))
    return out


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)"""
    out = ctypes.create_string_buffer(128)
    _check(lib().plk_comm_unique_id(out))
    return out.raw


def g1_to_bytes(p):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = ctypes.create_string_buffer(64)
    _check(lib().plk_g1_to_bytes(_np(p), out))
    return out.raw


def g1_from_bytes(b):
    out = np.zeros(8, dtype=np.uint64)
    _check(lib().plk_g1_from_bytes(bytes(b), _np(out)))
    return out


def fr_to_bytes(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = ctypes.create_string_buffer(32)
    _check(lib().plk_fr_to_bytes(_np(a), out))
    return out.raw


def fr_from_bytes(b):
    out = np.zeros(4, dtype=np.uint64)
    _check(lib().plk_fr_from_bytes(bytes(b), _np(out)))
    return out


def verify(vk_bytes, proof_bytes, strict_inputs=None):
    """plonk::verify(&vk, &proof, "keccak") (src/plonk.rs:189-210) on the bytes of vk.bin / proof.bin; pure CPU.
    strict_inputs: None = plk_verify (accepts num_inputs = 0 unless PLK_VERIFY_STRICT_INPUTS is set); True / False = plk_verify_ex with /
    without PLK_VERIFY_STRICT_INPUTS (the Solidity verifier's `num_inputs >= 1`, contrib/template.sol:697)."""
    valid = ctypes.c_int32(0)
    if strict_inputs is None:
        _check(lib().plk_verify(bytes(vk_bytes), ctypes.c_uint64(len(vk_bytes)), bytes(proof_bytes), ctypes.c_uint64(len(proof_bytes)), ctypes.byref(valid)))
    else:
        _check(lib().plk_verify_ex(bytes(vk_bytes), ctypes.c_uint64(len(vk_bytes)), bytes(proof_bytes), ctypes.c_uint64(len(proof_bytes)),
                                   ctypes.c_uint32(1 if strict_inputs else 0), ctypes.byref(valid)))
    return bool(valid.value)


def pairing_check(a, g2_a, b, g2_b):
    """e(a, g2_a) * e(b, g2_b) == 1 ; G1 as 8 x u64 Montgomery affine, G2 as the 128-byte file encoding"""
    out = ctypes.c_int32(0)
    a = np.ascontiguousarray(a, dtype=np.uint64); b = np.ascontiguousarray(b, dtype=np.uint64)
    _check(lib().plk_pairing_check(_np(a), bytes(g2_a), _np(b), bytes(g2_b), ctypes.byref(out)))
    return bool(out.value)


def crs42_g2_bytes():
    out = ctypes.create_string_buffer(256)
    lib().plk_crs42_g2_bytes(out)
    return out.raw


def keccak256(data):
    out = ctypes.create_string_buffer(32)
    lib().plk_keccak256(bytes(data), ctypes.c_uint64(len(data)), out)
    return out.raw


class _TranscriptStruct(ctypes.Structure):
    _fields_ = [("state0", ctypes.c_uint8 * 32), ("state1", ctypes.c_uint8 * 32), ("counter", ctypes.c_uint32)]


class Transcript:
    """RollingKeccakTranscript (src/plonk.rs:10; contrib/template.sol:267-307)."""

    def __init__(self):
        self._t = _TranscriptStruct()
        lib().plk_transcript_init(ctypes.byref(self._t))

    def absorb_fr(self, a):
        lib().plk_transcript_absorb_fr(ctypes.byref(self._t), _np(np.ascontiguousarray(a, dtype=np.uint64)))

    def absorb_g1(self, p):
        lib().plk_transcript_absorb_g1(ctypes.byref(self._t), _np(np.ascontiguousarray(p, dtype=np.uint64)))

    def challenge(self):
        out = np.zeros(4, dtype=np.uint64)
        lib().plk_transcript_challenge(ctypes.byref(self._t), _np(out))
        return out
