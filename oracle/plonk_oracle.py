"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the `plonkit prove` / `export-verification-key` / `verify` path, used as the
parity checker for the HIP product.  Protocol orchestration is Python; every O(N) vector step is
done by the plain-C library (oracle/c) through oracle_lib.

What it follows (reference file:line):
  loaders            src/reader.rs:92-175,178-241 ; src/r1cs_file.rs:44-154 ; src/circom_circuit.rs:15-38
  synthesize         src/circom_circuit.rs:74-133   (inputs 1..num_inputs, aux, LC fold, skip 0*LC=0)
  analyse            src/plonk.rs:72-93 ; golden string src/tests.rs:14
  prepare/setup      src/plonk.rs:97-119            (transpile -> setup(): 7 selectors + 4 sigmas)
  make_verification_key src/plonk.rs:122-124
  prove              src/plonk.rs:132-176           (keccak, monomial path = prove_by_steps)
  verify             src/plonk.rs:189-210 ; algorithm contrib/template.sol:445-494,496-586,588-689,691-758
  transcript         contrib/template.sol:267-307
  file formats       SURVEY.md Appendix A.1 (decoded from keys/setup/setup_2^10.key, vk.bin, proof.bin)
The arithmetic behind those call sites lives in bellman_ce 0.3.2 @ 5809cc16 (Cargo.lock:109-111),
which is NOT under /root/reference; the prover rounds are restated from SURVEY.md Appendix A.3/A.4.

PINNING: tests/test_oracle_golden.py checks this module byte-for-byte against the reference's own
golden vectors tests/golden/{vk.bin,proof.bin} (src/tests.rs:31-46,49-73), the analyse string
(src/tests.rs:14) and the r1cs sample (src/r1cs_file.rs:164-242).
UNPINNED (no reference fixture reaches it; implemented from recollection of bellman's adaptor):
linear combinations with more than 3 variables (d_next chains), constant*LC merges, quadratic gates.
"""
import json
import struct

import numpy as np

from . import oracle_lib as ol
from .oracle_lib import R_MOD, Q_MOD

NON_RESIDUES = (1, 5, 7, 10)       # coset representatives k_j of the 4 wire columns (vk.bin: 5,7,10)
COSET_GEN = 7                      # multiplicative generator used for the 4N coset
STATE_WIDTH = 4
AUX_OFFSET = 1                     # src/plonk.rs:24


# =============================================================================== loaders
class R1CS:
    def __init__(self, num_inputs, num_aux, num_variables, constraints):
        self.num_inputs, self.num_aux, self.num_variables = num_inputs, num_aux, num_variables
        self.constraints = constraints          # [(A, B, C)], each LC = [(wire, coeff_int)]


def load_r1cs_json(path_or_obj):
    """src/reader.rs:194-218: BTreeMap<String,String> => terms ordered by the *string* key."""
    cj = path_or_obj if isinstance(path_or_obj, dict) else json.load(open(path_or_obj))
    num_inputs = cj["nPubInputs"] + cj["nOutputs"] + 1
    conv = lambda lc: [(int(k), int(lc[k]) % R_MOD) for k in sorted(lc.keys())]
    cons = [(conv(c[0]), conv(c[1]), conv(c[2])) for c in cj["constraints"]]
    return R1CS(num_inputs, cj["nVars"] - num_inputs, cj["nVars"], cons)


BN254_PRIME_LE = bytes.fromhex("010000f093f5e1439170b97948e833285d588181b64550b829a031e1724e6430")


def parse_r1cs_bin(data: bytes):
    """src/r1cs_file.rs:100-154.  Returns (header dict, constraints, wire_mapping)."""
    try:
        return _parse_r1cs_bin(data)
    except (struct.error, KeyError, IndexError) as e:      # truncated / inconsistent section table
        raise ValueError("InvalidData: %s" % e)


def _parse_r1cs_bin(data: bytes):
    if data[:4] != b"r1cs":
        raise ValueError("Invalid magic number")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version != 1:
        raise ValueError("Unsupported version")
    off, secs = 12, {}
    for _ in range(nsec):
        t, sz = struct.unpack_from("<IQ", data, off)
        off += 12
        secs[t] = (off, sz)
        off += sz
    o, sz = secs[1]
    (field_size,) = struct.unpack_from("<I", data, o)
    prime = data[o + 4:o + 4 + field_size]
    if sz != 32 + field_size:
        raise ValueError("Invalid header section size")
    n_wires, n_pub_out, n_pub_in, n_prv_in, n_labels, n_constraints = struct.unpack_from("<IIIIQI", data, o + 4 + field_size)
    if field_size != 32:
        raise ValueError("This parser only supports 32-byte fields")
    if prime != BN254_PRIME_LE:
        raise ValueError("This parser only supports bn256")
    hdr = dict(field_size=field_size, n_wires=n_wires, n_pub_out=n_pub_out, n_pub_in=n_pub_in,
               n_prv_in=n_prv_in, n_labels=n_labels, n_constraints=n_constraints)
    o, _ = secs[2]
    cons = []
    for _ in range(n_constraints):
        abc = []
        for _ in range(3):
            (nv,) = struct.unpack_from("<I", data, o)
            o += 4
            lc = []
            for _ in range(nv):
                (w,) = struct.unpack_from("<I", data, o)
                v = int.from_bytes(data[o + 4:o + 36], "little")
                if v >= R_MOD:
                    raise ValueError("not in field")
                lc.append((w, v))
                o += 36
            abc.append(lc)
        cons.append(tuple(abc))
    o, sz = secs[3]
    if sz != n_wires * 8:
        raise ValueError("Invalid map section size")
    wmap = list(struct.unpack_from("<%dQ" % n_wires, data, o))
    if wmap[0] != 0:
        raise ValueError("Wire 0 should always be mapped to 0")
    return hdr, cons, wmap


def load_r1cs_bin(data: bytes):
    """src/reader.rs:227-241 (the wire map is dropped, src/reader.rs:182)."""
    hdr, cons, _ = parse_r1cs_bin(data)
    num_inputs = 1 + hdr["n_pub_in"] + hdr["n_pub_out"]
    return R1CS(num_inputs, hdr["n_wires"] - num_inputs, hdr["n_wires"], cons)


class R1CSFlat:
    """the same R1CS held flat (all linear combinations back to back + one offset table), parsed by the C front end of
    oracle/c/oracle.c: what bench.py's cpu_baseline leg proves from, so that its prove is compiled code end to end"""
    def __init__(self, num_inputs, num_variables, off, wires, coeffs):
        self.num_inputs, self.num_variables, self.num_aux = num_inputs, num_variables, num_variables - num_inputs
        self.off, self.wires, self.coeffs = off, wires, coeffs

    @property
    def num_constraints(self):
        return (self.off.shape[0] - 1) // 3


def load_r1cs_flat(data: bytes) -> R1CSFlat:
    hdr, off, wires, coeffs = ol.r1cs_parse(data)
    return R1CSFlat(1 + hdr["n_pub_in"] + hdr["n_pub_out"], hdr["n_wires"], off, wires, coeffs)


def load_witness_json(path):
    return [int(x) % R_MOD for x in json.load(open(path))]


def parse_wtns(data: bytes):
    """src/reader.rs:124-175."""
    if data[:4] != b"wtns":
        raise ValueError("invalid file header")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version > 2:
        raise ValueError("unsupported file version")
    if nsec != 2:
        raise ValueError("invalid num sections")
    t, sz = struct.unpack_from("<IQ", data, 12)
    if t != 1:
        raise ValueError("invalid section type")
    if sz != 40:
        raise ValueError("invalid section len")
    (fs,) = struct.unpack_from("<I", data, 24)
    if fs != 32:
        raise ValueError("invalid field byte size")
    if data[28:60] != BN254_PRIME_LE:
        raise ValueError("invalid curve prime")
    (wl,) = struct.unpack_from("<I", data, 60)
    t, sz = struct.unpack_from("<IQ", data, 64)
    if t != 2:
        raise ValueError("invalid section type")
    if sz != wl * 32:
        raise ValueError("invalid witness section size")
    out = []
    for i in range(wl):
        v = int.from_bytes(data[76 + 32 * i:108 + 32 * i], "little")
        if v >= R_MOD:
            raise ValueError("not in field")
        out.append(v)
    return out


# ------------------------------------------------------------------------- SRS container
G2_BYTES = 128


class Crs:
    """Crs<E, CrsForMonomialForm|CrsForLagrangeForm>: g1 = Montgomery affine [n,8]."""
    def __init__(self, g1, g2_raw):
        self.g1, self.g2_raw = g1, g2_raw       # g2 kept as its 2x128 file bytes (only re-serialised)


def g1_from_bytes(b):
    if b[0] & 0x40:
        return np.zeros(8, dtype=np.uint64)
    return ol.g1_from_ints(int.from_bytes(b[:32], "big"), int.from_bytes(b[32:64], "big"))


def g1_to_bytes(p):
    if ol.g1_is_inf(p):
        return b"\x40" + b"\x00" * 63
    x, y = ol.g1_to_ints(p)
    return x.to_bytes(32, "big") + y.to_bytes(32, "big")


def read_crs(data: bytes):
    (n1,) = struct.unpack_from(">Q", data, 0)
    raw = np.frombuffer(data, dtype=np.uint8, count=64 * n1, offset=8).reshape(n1, 2, 32)
    # big-endian canonical -> canonical LE limbs -> Montgomery
    le = raw[:, :, ::-1].copy().view("<u8").reshape(n1 * 2, 4)
    inf = (raw[:, 0, 0] & 0x40) != 0
    mont = np.zeros_like(le)
    ol.lib().orc_fq_from_canonical(ol._p(mont), ol._p(np.ascontiguousarray(le)), 2 * n1)
    g1 = mont.reshape(n1, 8)
    g1[inf] = 0
    off = 8 + 64 * n1
    (n2,) = struct.unpack_from(">Q", data, off)
    assert n2 == 2
    return Crs(g1, data[off + 8:off + 8 + 2 * G2_BYTES])


def write_crs(crs: Crs) -> bytes:
    n1 = crs.g1.shape[0]
    canon = np.zeros((n1 * 2, 4), dtype=np.uint64)
    ol.lib().orc_fq_to_canonical(ol._p(canon), ol._p(np.ascontiguousarray(crs.g1.reshape(n1 * 2, 4))), 2 * n1)
    be = canon.view(np.uint8).reshape(n1, 2, 32)[:, :, ::-1].copy()
    inf = ~np.any(crs.g1.reshape(n1, 8), axis=1)
    be[inf] = 0
    be[inf, 0, 0] = 0x40
    return struct.pack(">Q", n1) + be.tobytes() + struct.pack(">Q", 2) + crs.g2_raw


# ============================================================================ transpiler
class Gate:
    __slots__ = ("vars", "coeffs")

    def __init__(self, vars_, coeffs):
        self.vars, self.coeffs = vars_, coeffs     # 4 var ids ; [q_a,q_b,q_c,q_d,q_m,q_const,q_d_next]


class Transpiled:
    """Variables: id 0 = dummy (value 0); id w (1 <= w < num_variables) = circom wire w
    (src/circom_circuit.rs:107-113 with AUX_OFFSET = 1); ids >= num_variables = transpiler temporaries."""
    def __init__(self):
        self.gates, self.values, self.stats, self.num_hints = [], None, [], 0


def _split_lc(lc):
    """stable de-duplication; wire 0 (ONE) goes to the constant term."""
    const, order, acc = 0, [], {}
    for w, c in lc:
        if w == 0:
            const = (const + c) % R_MOD
        elif w in acc:
            acc[w] = (acc[w] + c) % R_MOD
        else:
            acc[w] = c % R_MOD
            order.append(w)
    return const, [(w, acc[w]) for w in order if acc[w] != 0]


def transpile(r1cs: R1CS, witness=None) -> Transpiled:
    T = Transpiled()
    vals = None
    if witness is not None:
        vals = [0] + [v % R_MOD for v in witness[1:r1cs.num_variables]]
    T.values = vals
    next_id = [r1cs.num_variables]

    def alloc(v):
        i = next_id[0]
        next_id[0] += 1
        if vals is not None:
            vals.append(v % R_MOD)
        return i

    def val(i):
        return vals[i] if vals is not None else 0

    def new_gate(vs, cs):
        T.gates.append(Gate(list(vs), [c % R_MOD for c in cs]))

    def lc_as_gates(lc, mult, free, collapse):
        """returns (var or None, coeff).  [recollection of bellman adaptor::enforce_lc_as_gates]"""
        assert len(lc) > 0
        if len(lc) == 1 and free == 0 and collapse:
            return lc[0][0], lc[0][1]
        if mult != 1:
            lc = [(w, c * mult % R_MOD) for w, c in lc]
            free = free * mult % R_MOD
        final = None
        if collapse:
            v = (sum(c * val(w) for w, c in lc) + free) % R_MOD
            final = alloc(v)
            lc = lc + [(final, R_MOD - 1)]
        if len(lc) <= STATE_WIDTH:
            vs = [w for w, _ in lc] + [0] * (STATE_WIDTH - len(lc))
            cs = [c for _, c in lc] + [0] * (STATE_WIDTH - len(lc))
            new_gate(vs, cs + [0, free, 0])
        else:                                   # UNPINNED: d_next chain
            it = list(lc)
            head, it = it[:STATE_WIDTH], it[STATE_WIDTH:]
            s = (sum(c * val(w) for w, c in head) + free) % R_MOD
            nxt = alloc(s)
            new_gate([w for w, _ in head], [c for _, c in head] + [0, free, R_MOD - 1])
            while len(it) > STATE_WIDTH - 1:
                chunk, it = it[:STATE_WIDTH - 1], it[STATE_WIDTH - 1:]
                s = (sum(c * val(w) for w, c in chunk) + val(nxt)) % R_MOD
                nn = alloc(s)
                pad = STATE_WIDTH - 1 - len(chunk)
                new_gate([w for w, _ in chunk] + [0] * pad + [nxt], [c for _, c in chunk] + [0] * pad + [1, 0, 0, R_MOD - 1])
                nxt = nn
            pad = STATE_WIDTH - 1 - len(it)
            new_gate([w for w, _ in it] + [0] * pad + [nxt], [c for _, c in it] + [0] * pad + [1, 0, 0, 0])
        return final, 1

    for idx, (A, B, C) in enumerate(r1cs.constraints):
        if (len(A) == 0 or len(B) == 0) and len(C) == 0:       # src/circom_circuit.rs:121-122
            continue
        g0 = len(T.gates)
        ac, al = _split_lc(A)
        bc, bl = _split_lc(B)
        cc, cl = _split_lc(C)
        a_k, b_k, c_k = len(al) == 0, len(bl) == 0, len(cl) == 0
        if a_k and b_k:
            free = (cc - ac * bc) % R_MOD
            if c_k:
                assert free == 0, "unsatisfiable constant constraint"
            else:
                lc_as_gates(cl, 1, free, False)
        elif a_k or b_k:                        # UNPINNED: constant * LC = LC
            k, lin, lin_c = (ac, bl, bc) if a_k else (bc, al, ac)
            merged = [(w, c * k % R_MOD) for w, c in lin] + [(w, (R_MOD - c) % R_MOD) for w, c in cl]
            free = (k * lin_c - cc) % R_MOD
            _, merged = _split_lc(merged)
            if merged:
                lc_as_gates(merged, 1, free, False)
            else:
                assert free == 0
        else:
            same = len(al) == 1 and len(bl) == 1 and al[0][0] == bl[0][0] and (c_k or (len(cl) == 1 and cl[0][0] == al[0][0]))
            if same:                            # UNPINNED: quadratic gate on one variable
                x, a1, b1 = al[0][0], al[0][1], bl[0][1]
                c1 = 0 if c_k else cl[0][1]
                new_gate([x, x, 0, 0], [(ac * b1 + a1 * bc - c1) % R_MOD, 0, 0, 0, a1 * b1 % R_MOD, (ac * bc - cc) % R_MOD, 0])
            else:
                av, acoef = lc_as_gates(al, 1, ac, True)
                bv, bcoef = lc_as_gates(bl, 1, bc, True)
                if c_k:
                    new_gate([av, bv, 0, 0], [0, 0, 0, 0, acoef * bcoef % R_MOD, (R_MOD - cc) % R_MOD, 0])
                else:
                    cv, ccoef = lc_as_gates(cl, 1, cc, True)
                    new_gate([av, bv, cv, 0], [0, 0, (R_MOD - ccoef) % R_MOD, 0, acoef * bcoef % R_MOD, 0, 0])
        T.stats.append({"name": str(idx), "num_gates": len(T.gates) - g0})
        T.num_hints += 1
    return T


def analyse(r1cs: R1CS) -> str:
    """src/plonk.rs:72-93, serialised like serde_json::to_string (src/tests.rs:14)."""
    T = transpile(r1cs)
    d = {"num_inputs": r1cs.num_inputs, "num_aux": r1cs.num_aux, "num_variables": r1cs.num_variables,
         "num_constraints": len(r1cs.constraints), "num_nontrivial_constraints": len(T.stats),
         "num_gates": len(T.gates), "num_hints": T.num_hints}
    if T.stats:
        d["constraint_stats"] = T.stats
    return json.dumps(d, separators=(",", ":"))


# ================================================================================= setup
class Setup:
    """SetupPolynomials: n gates (n+1 = domain size N), selectors and sigmas in coefficient form."""
    pass


def _assemble(r1cs: R1CS, T: Transpiled):
    n_in = r1cs.num_inputs - 1
    rows = [Gate([i, 0, 0, 0], [R_MOD - 1, 0, 0, 0, 0, 0, 0]) for i in range(1, n_in + 1)] + T.gates
    n_real = len(rows)
    N = 1
    while N < n_real + 1:
        N *= 2
    return rows, n_in, N


def setup(r1cs: R1CS, T: Transpiled = None) -> Setup:
    T = T or transpile(r1cs)
    rows, n_in, N = _assemble(r1cs, T)
    log_n = N.bit_length() - 1
    S = Setup()
    S.n, S.N, S.log_n, S.num_inputs = N - 1, N, log_n, n_in
    sel = [[0] * N for _ in range(7)]
    for r, g in enumerate(rows):
        for k in range(7):
            sel[k][r] = g.coeffs[k]
    w = ol.omega(log_n)
    dom = [1] * N
    for i in range(1, N):
        dom[i] = dom[i - 1] * w % R_MOD
    sig = [[NON_RESIDUES[j] * dom[i] % R_MOD for i in range(N)] for j in range(4)]
    occ = {}
    for r, g in enumerate(rows):
        for j in range(4):
            v = g.vars[j]
            if v != 0:
                occ.setdefault(v, []).append((j, r))
    for lst in occ.values():
        if len(lst) > 1:
            for k, (j, r) in enumerate(lst):
                j2, r2 = lst[(k + 1) % len(lst)]
                sig[j][r] = NON_RESIDUES[j2] * dom[r2] % R_MOD
    S.selector_values = [ol.fr_vec(s) for s in sel]
    S.sigma_values = [ol.fr_vec(s) for s in sig]
    S.selectors = [ol.ntt(v, log_n, inverse=True) for v in S.selector_values]     # 7 x iNTT(N)
    S.sigmas = [ol.ntt(v, log_n, inverse=True) for v in S.sigma_values]           # 4 x iNTT(N)
    return S


def _rows_flat(rf: R1CSFlat, witness=None):
    """public-input rows + the C transpiler's gates, padded: vars uint32 [4, N], q [7, N, 4] (Montgomery), values"""
    n_in = rf.num_inputs - 1
    vars_, q, g, values, _ = ol.transpile_c(rf.off, rf.wires, rf.coeffs, rf.num_variables, witness, first_row=n_in)
    n_real = n_in + g
    N = 1
    while N < n_real + 1:
        N *= 2
    V = np.zeros((4, N), dtype=np.uint32)
    V[:, :n_real] = vars_[:, :n_real]
    V[0, :n_in] = np.arange(1, n_in + 1, dtype=np.uint32)
    V[1:, :n_in] = 0
    Q = np.zeros((7, N, 4), dtype=np.uint64)
    Q[:, :n_real] = q[:, :n_real]
    Q[:, :n_in] = 0
    Q[0, :n_in] = ol.fr_mont(R_MOD - 1)
    return V, Q, n_in, N, values


def setup_flat(rf: R1CSFlat) -> Setup:
    """setup() from the flat form: gates by the C transpiler, the permutation by one stable sort (numpy) — same
    polynomials as setup(load_r1cs_bin(..)) (tests/test_oracle_golden.py)"""
    V, Q, n_in, N, _ = _rows_flat(rf)
    log_n = N.bit_length() - 1
    S = Setup()
    S.n, S.N, S.log_n, S.num_inputs = N - 1, N, log_n, n_in
    S.selector_values = [np.ascontiguousarray(Q[k]) for k in range(7)]
    dom = ol.vpowers(ol.omega(log_n), N)
    kdom = np.stack([ol.vscale(dom, NON_RESIDUES[j]) for j in range(4)])          # [4, N, 4]: k_j * omega^r
    # occurrences in row-major order (gate by gate, a -> d inside a gate): slot index r * 4 + j
    flat = np.ascontiguousarray(V.T).reshape(-1)                                 # [N * 4], value = variable id
    order = np.argsort(flat, kind="stable")
    sv = flat[order]
    nxt = np.arange(4 * N, dtype=np.int64)                                       # identity
    same_next = np.empty(4 * N, dtype=bool); same_next[:-1] = sv[1:] == sv[:-1]; same_next[-1] = False
    first = np.empty(4 * N, dtype=bool); first[0] = True; first[1:] = sv[1:] != sv[:-1]
    # group start of every sorted position
    starts = np.maximum.accumulate(np.where(first, np.arange(4 * N), 0))
    succ = np.where(same_next, np.roll(order, -1), order[starts])                # next occurrence, the last wraps to the first
    live = sv != 0
    nxt[order[live]] = succ[live]
    jj, rr = nxt % 4, nxt // 4                                                   # successor slot of slot (r * 4 + j)
    sig = kdom[jj, rr].reshape(N, 4, 4)                                          # [r, j, limbs]
    S.sigma_values = [np.ascontiguousarray(sig[:, j]) for j in range(4)]
    S.selectors = [ol.ntt(v, log_n, inverse=True) for v in S.selector_values]
    S.sigmas = [ol.ntt(v, log_n, inverse=True) for v in S.sigma_values]
    return S


def commit(crs: Crs, coeffs):
    n = coeffs.shape[0]
    assert crs.g1.shape[0] >= n, "SRS too small"
    return ol.msm(crs.g1[:n], coeffs)


class VerificationKey:
    pass


def make_verification_key(S: Setup, crs: Crs) -> VerificationKey:
    vk = VerificationKey()
    vk.n, vk.num_inputs = S.n, S.num_inputs
    vk.selector_commitments = [commit(crs, S.selectors[k]) for k in range(6)]
    vk.next_step_selector_commitments = [commit(crs, S.selectors[6])]
    vk.permutation_commitments = [commit(crs, s) for s in S.sigmas]
    vk.non_residues = list(NON_RESIDUES[1:])
    vk.g2_raw = crs.g2_raw
    return vk


def write_vk(vk) -> bytes:
    b = struct.pack(">QQ", vk.n, vk.num_inputs)
    b += struct.pack(">Q", 6) + b"".join(g1_to_bytes(p) for p in vk.selector_commitments)
    b += struct.pack(">Q", 1) + b"".join(g1_to_bytes(p) for p in vk.next_step_selector_commitments)
    b += struct.pack(">Q", 4) + b"".join(g1_to_bytes(p) for p in vk.permutation_commitments)
    b += struct.pack(">Q", 3) + b"".join(k.to_bytes(32, "big") for k in vk.non_residues)
    return b + vk.g2_raw


def read_vk(data: bytes) -> VerificationKey:
    vk = VerificationKey()
    vk.n, vk.num_inputs = struct.unpack_from(">QQ", data, 0)
    o = 16

    def pts(o):
        (k,) = struct.unpack_from(">Q", data, o)
        o += 8
        out = [g1_from_bytes(data[o + 64 * i:o + 64 * i + 64]) for i in range(k)]
        return out, o + 64 * k
    vk.selector_commitments, o = pts(o)
    vk.next_step_selector_commitments, o = pts(o)
    vk.permutation_commitments, o = pts(o)
    (k,) = struct.unpack_from(">Q", data, o)
    o += 8
    vk.non_residues = [int.from_bytes(data[o + 32 * i:o + 32 * i + 32], "big") for i in range(k)]
    o += 32 * k
    vk.g2_raw = data[o:o + 256]
    return vk


# ============================================================================ transcript
class Transcript:
    """RollingKeccakTranscript (contrib/template.sol:267-307)."""
    def __init__(self):
        self.s0 = self.s1 = b"\x00" * 32
        self.counter = 0

    def absorb_u256(self, v: int):
        w = v.to_bytes(32, "big")
        o0, o1 = self.s0, self.s1
        self.s0 = ol.keccak256(struct.pack(">I", 0) + o0 + o1 + w)
        self.s1 = ol.keccak256(struct.pack(">I", 1) + o0 + o1 + w)

    def absorb_fr(self, v):
        self.absorb_u256(v)

    def absorb_g1(self, p):
        x, y = (0, 0) if ol.g1_is_inf(p) else ol.g1_to_ints(p)
        self.absorb_u256(x)
        self.absorb_u256(y)

    def challenge(self) -> int:
        q = ol.keccak256(struct.pack(">I", 2) + self.s0 + self.s1 + struct.pack(">I", self.counter))
        self.counter += 1
        return int.from_bytes(q, "big") & ((1 << 253) - 1)


# ================================================================================ prover
class Proof:
    pass


def is_satisfied(r1cs, T, S, rows=None):
    """is_satisfied_using_one_shot_check (src/plonk.rs:128,137): every gate equation holds."""
    rows = rows or _assemble(r1cs, T)[0]
    v = T.values
    for r, g in enumerate(rows):
        a, b, c, d = (v[x] for x in g.vars)
        q = g.coeffs
        dn = v[rows[r + 1].vars[3]] if r + 1 < len(rows) else 0
        pi = v[g.vars[0]] if r < S.num_inputs else 0
        if (q[0] * a + q[1] * b + q[2] * c + q[3] * d + q[4] * a * b + q[5] + q[6] * dn + pi) % R_MOD:
            return False
    return True


def prove(r1cs: R1CS, witness, crs: Crs, S: Setup = None, return_debug=False) -> Proof:
    """prove_by_steps with RollingKeccakTranscript and the monomial-form key only
    (src/plonk.rs:152-159) — rounds per SURVEY.md Appendix A.4; no blinding."""
    if isinstance(r1cs, R1CSFlat):
        # compiled front end (oracle/c/oracle.c): synthesis with the witness (a Montgomery array, ol.wtns_parse), gate check
        S = S or setup_flat(r1cs)
        V, _, n_in, N, values = _rows_flat(r1cs, witness)
        assert N == S.N
        cols = ol.gather_columns(V, values, N)
        assert ol.check_gates(cols, np.stack(S.selector_values), N, n_in), "must satisfy"
        inputs = ol.fr_ints(values[1:n_in + 1]) if n_in else []
        w_vals = [cols[j] for j in range(4)]
    else:
        T = transpile(r1cs, witness)
        S = S or setup(r1cs, T)
        rows, n_in, N = _assemble(r1cs, T)
        assert N == S.N
        assert is_satisfied(r1cs, T, S, rows), "must satisfy"
        vals = T.values
        inputs = [vals[i] for i in range(1, n_in + 1)]
        cols = [[0] * N for _ in range(4)]
        for r, g in enumerate(rows):
            for j in range(4):
                cols[j][r] = vals[g.vars[j]]
        w_vals = [ol.fr_vec(c) for c in cols]
    log_n, log_4n = S.log_n, S.log_n + 2
    w_omega = ol.omega(log_n)

    # ---- round 1: wire polynomials
    w_coef = [ol.ntt(v, log_n, inverse=True) for v in w_vals]                      # 4 x iNTT(N)
    P = Proof()
    P.n, P.inputs = S.n, inputs
    P.wire_commitments = [commit(crs, c) for c in w_coef]                          # 4 x MSM(N)
    tr = Transcript()
    for x in inputs:
        tr.absorb_fr(x)
    for c in P.wire_commitments:
        tr.absorb_g1(c)
    beta, gamma = tr.challenge(), tr.challenge()

    # ---- round 2: grand product z
    dom = ol.vpowers(w_omega, N)
    num = den = None
    for j in range(4):
        nj = ol.vadd_scalar(ol.vaxpy(w_vals[j], beta * NON_RESIDUES[j] % R_MOD, dom), gamma)
        dj = ol.vadd_scalar(ol.vaxpy(w_vals[j], beta, S.sigma_values[j]), gamma)
        num = nj if num is None else ol.vmul(num, nj)
        den = dj if den is None else ol.vmul(den, dj)
    ratio = ol.vmul(num, ol.vbatch_inv(den))
    z_vals = ol.vshifted_prefix_product(ratio)                                     # z_0 = 1, N values
    z_coef = ol.ntt(z_vals, log_n, inverse=True)                                   # iNTT(N)
    P.grand_product_commitment = commit(crs, z_coef)                               # MSM(N)
    tr.absorb_g1(P.grand_product_commitment)
    alpha = tr.challenge()

    # ---- round 3: quotient on the coset 7*<w_4N>
    M = 4 * N

    def lde(coef):
        ext = ol.fr_zeros(M)
        ext[:N] = coef
        return ol.ntt(ext, log_4n, coset=COSET_GEN)
    w_e = [lde(c) for c in w_coef]
    z_e = lde(z_coef)
    q_e = [lde(c) for c in S.selectors]
    s_e = [lde(c) for c in S.sigmas]
    pi_vals = ol.fr_zeros(N)
    if inputs:
        pi_vals[:len(inputs)] = ol.fr_vec(inputs)
    pi_e = lde(ol.ntt(pi_vals, log_n, inverse=True))
    l0_vals = ol.fr_zeros(N)
    l0_vals[0] = ol.fr_mont(1)
    l0_e = lde(ol.ntt(l0_vals, log_n, inverse=True))
    x_e = ol.vpowers(ol.omega(log_4n), M, COSET_GEN)                               # the coset points
    shift = lambda v: np.roll(v, -4, axis=0)                                       # f(w x) on the 4N domain
    gate = q_e[5]
    for j in range(4):
        gate = ol.vadd(gate, ol.vmul(q_e[j], w_e[j]))
    gate = ol.vadd(gate, ol.vmul(q_e[4], ol.vmul(w_e[0], w_e[1])))
    gate = ol.vadd(gate, ol.vmul(q_e[6], shift(w_e[3])))
    gate = ol.vadd(gate, pi_e)
    pa = z_e
    pb = shift(z_e)
    for j in range(4):
        pa = ol.vmul(pa, ol.vadd_scalar(ol.vaxpy(w_e[j], beta * NON_RESIDUES[j] % R_MOD, x_e), gamma))
        pb = ol.vmul(pb, ol.vadd_scalar(ol.vaxpy(w_e[j], beta, s_e[j]), gamma))
    perm = ol.vsub(pa, pb)
    l0_part = ol.vmul(l0_e, ol.vadd_scalar(z_e, R_MOD - 1))
    tnum = ol.vaxpy(ol.vaxpy(gate, alpha, perm), alpha * alpha % R_MOD, l0_part)
    zh = ol.vadd_scalar(ol.vpowers(pow(ol.omega(log_4n), N, R_MOD), M, pow(COSET_GEN, N, R_MOD)), R_MOD - 1)
    t_e = ol.vmul(tnum, ol.vbatch_inv(zh))
    t_coef = ol.ntt(t_e, log_4n, inverse=True, coset=COSET_GEN)                    # coset-iNTT(4N)
    t_parts = [np.ascontiguousarray(t_coef[k * N:(k + 1) * N]) for k in range(4)]
    P.quotient_poly_commitments = [commit(crs, t) for t in t_parts]                # 4 x MSM(N)
    for c in P.quotient_poly_commitments:
        tr.absorb_g1(c)
    z = tr.challenge()

    # ---- round 4: evaluations + linearisation
    zw = z * w_omega % R_MOD
    P.wire_values_at_z = [ol.poly_eval(c, z) for c in w_coef]
    P.wire_values_at_z_omega = [ol.poly_eval(w_coef[3], zw)]
    P.permutation_polynomials_at_z = [ol.poly_eval(S.sigmas[j], z) for j in range(3)]
    P.quotient_polynomial_at_z = ol.poly_eval(t_coef, z)
    P.grand_product_at_z_omega = ol.poly_eval(z_coef, zw)
    wz = P.wire_values_at_z
    zN = pow(z, N, R_MOD)
    l0_z = (zN - 1) * pow(N * (z - 1) % R_MOD, -1, R_MOD) % R_MOD
    r = S.selectors[5].copy()
    for j in range(4):
        r = ol.vaxpy(r, wz[j], S.selectors[j])
    r = ol.vaxpy(r, wz[0] * wz[1] % R_MOD, S.selectors[4])
    r = ol.vaxpy(r, P.wire_values_at_z_omega[0], S.selectors[6])
    fz = alpha
    for j in range(4):
        fz = fz * ((wz[j] + beta * NON_RESIDUES[j] * z + gamma) % R_MOD) % R_MOD
    fz = (fz + alpha * alpha * l0_z) % R_MOD
    r = ol.vaxpy(r, fz, z_coef)
    fs = alpha * beta * P.grand_product_at_z_omega % R_MOD
    for j in range(3):
        fs = fs * ((wz[j] + beta * P.permutation_polynomials_at_z[j] + gamma) % R_MOD) % R_MOD
    r = ol.vaxpy(r, (R_MOD - fs) % R_MOD, S.sigmas[3])
    P.linearization_polynomial_at_z = ol.poly_eval(r, z)
    for v in wz + P.wire_values_at_z_omega + P.permutation_polynomials_at_z:
        tr.absorb_fr(v)
    tr.absorb_fr(P.quotient_polynomial_at_z)
    tr.absorb_fr(P.linearization_polynomial_at_z)
    tr.absorb_fr(P.grand_product_at_z_omega)
    v = tr.challenge()

    # ---- round 5: openings
    agg = t_parts[0].copy()
    zp = 1
    for k in range(1, 4):
        zp = zp * zN % R_MOD
        agg = ol.vaxpy(agg, zp, t_parts[k])
    vp = v
    agg = ol.vaxpy(agg, vp, r)
    for poly in w_coef + S.sigmas[:3]:
        vp = vp * v % R_MOD
        agg = ol.vaxpy(agg, vp, poly)
    W_z = ol.poly_div_linear(agg, z)
    vp = vp * v % R_MOD
    agg2 = ol.vscale(z_coef, vp)
    vp = vp * v % R_MOD
    agg2 = ol.vaxpy(agg2, vp, w_coef[3])
    W_zw = ol.poly_div_linear(agg2, zw)
    P.opening_at_z_proof = commit(crs, W_z)                                        # 2 x MSM(N)
    P.opening_at_z_omega_proof = commit(crs, W_zw)
    if return_debug:
        return P, dict(beta=beta, gamma=gamma, alpha=alpha, z=z, v=v, w_coef=w_coef, z_coef=z_coef, z_vals=z_vals,
                       t_coef=t_coef, r=r, W_z=W_z, W_zw=W_zw, w_vals=w_vals, setup=S)
    return P


def write_proof(P) -> bytes:
    fr = lambda x: x.to_bytes(32, "big")
    b = struct.pack(">QQ", P.n, len(P.inputs)) + b"".join(fr(x) for x in P.inputs)
    b += struct.pack(">Q", 4) + b"".join(g1_to_bytes(c) for c in P.wire_commitments)
    b += g1_to_bytes(P.grand_product_commitment)
    b += struct.pack(">Q", 4) + b"".join(g1_to_bytes(c) for c in P.quotient_poly_commitments)
    b += struct.pack(">Q", 4) + b"".join(fr(x) for x in P.wire_values_at_z)
    b += struct.pack(">Q", 1) + b"".join(fr(x) for x in P.wire_values_at_z_omega)
    b += fr(P.grand_product_at_z_omega) + fr(P.quotient_polynomial_at_z) + fr(P.linearization_polynomial_at_z)
    b += struct.pack(">Q", 3) + b"".join(fr(x) for x in P.permutation_polynomials_at_z)
    return b + g1_to_bytes(P.opening_at_z_proof) + g1_to_bytes(P.opening_at_z_omega_proof)


def read_proof(data: bytes) -> Proof:
    P = Proof()
    o = [0]

    def u64():
        (v,) = struct.unpack_from(">Q", data, o[0])
        o[0] += 8
        return v

    def fr():
        v = int.from_bytes(data[o[0]:o[0] + 32], "big")
        o[0] += 32
        return v

    def g1():
        p = g1_from_bytes(data[o[0]:o[0] + 64])
        o[0] += 64
        return p
    P.n = u64()
    P.inputs = [fr() for _ in range(u64())]
    P.wire_commitments = [g1() for _ in range(u64())]
    P.grand_product_commitment = g1()
    P.quotient_poly_commitments = [g1() for _ in range(u64())]
    P.wire_values_at_z = [fr() for _ in range(u64())]
    P.wire_values_at_z_omega = [fr() for _ in range(u64())]
    P.grand_product_at_z_omega, P.quotient_polynomial_at_z, P.linearization_polynomial_at_z = fr(), fr(), fr()
    P.permutation_polynomials_at_z = [fr() for _ in range(u64())]
    P.opening_at_z_proof, P.opening_at_z_omega_proof = g1(), g1()
    assert o[0] == len(data)
    return P


# ============================================================================== verifier
def verify(vk, P, tau=42):
    """contrib/template.sol verify_initial/verify_at_z/reconstruct_d/verify_commitments.  The final
    pairing product e(A, g2) * e(B, tau*g2) == 1 is checked through the trapdoor identity
    A + tau*B == O, valid only for the insecure crs_42 keys (tau = 42, src/plonk.rs:41,47)."""
    N = vk.n + 1
    log_n = N.bit_length() - 1
    om = ol.omega(log_n)
    # template.sol:697 also requires num_inputs >= 1; that is the Solidity verifier's restriction — the Rust verifier
    # `plonkit verify` calls has none [recollection, unpinned], and circom circuits without public signals are legal
    if len(P.inputs) != vk.num_inputs:
        return False
    tr = Transcript()
    for x in P.inputs:
        tr.absorb_fr(x)
    for c in P.wire_commitments:
        tr.absorb_g1(c)
    beta, gamma = tr.challenge(), tr.challenge()
    tr.absorb_g1(P.grand_product_commitment)
    alpha = tr.challenge()
    for c in P.quotient_poly_commitments:
        tr.absorb_g1(c)
    z = tr.challenge()
    zN = pow(z, N, R_MOD)
    if zN == 1:
        return False
    lag = [pow(om, i, R_MOD) * (zN - 1) % R_MOD * pow(N * (z - pow(om, i, R_MOD)) % R_MOD, -1, R_MOD) % R_MOD
           for i in range(max(vk.num_inputs, 1))]
    wz, sz = P.wire_values_at_z, P.permutation_polynomials_at_z
    lhs = (zN - 1) * P.quotient_polynomial_at_z % R_MOD
    rhs = P.linearization_polynomial_at_z
    for i, x in enumerate(P.inputs):
        rhs = (rhs + lag[i] * x) % R_MOD
    zpart = P.grand_product_at_z_omega
    for j in range(3):
        zpart = zpart * ((sz[j] * beta + gamma + wz[j]) % R_MOD) % R_MOD
    zpart = zpart * ((gamma + wz[3]) % R_MOD) % R_MOD * alpha % R_MOD
    rhs = (rhs - zpart - lag[0] * alpha * alpha) % R_MOD
    if lhs != rhs:
        return False
    for x in wz + P.wire_values_at_z_omega + sz:
        tr.absorb_fr(x)
    tr.absorb_fr(P.quotient_polynomial_at_z)
    tr.absorb_fr(P.linearization_polynomial_at_z)
    tr.absorb_fr(P.grand_product_at_z_omega)
    v = tr.challenge()
    tr.absorb_g1(P.opening_at_z_proof)
    tr.absorb_g1(P.opening_at_z_omega_proof)
    u = tr.challenge()

    add, mul, neg = ol.g1_add, ol.g1_mul, ol.g1_neg
    sel = vk.selector_commitments
    d = sel[5]
    for j in range(4):
        d = add(d, mul(sel[j], wz[j]))
    d = add(d, mul(sel[4], wz[0] * wz[1] % R_MOD))
    d = add(d, mul(vk.next_step_selector_commitments[0], P.wire_values_at_z_omega[0]))
    gz = (z * beta + wz[0] + gamma) % R_MOD
    for j in range(3):
        gz = gz * ((z * vk.non_residues[j] * beta + gamma + wz[j + 1]) % R_MOD) % R_MOD
    gz = (gz * alpha + lag[0] * alpha * alpha) % R_MOD
    gzw = pow(v, 1 + 1 + 4 + 4 - 1, R_MOD) * u % R_MOD
    last = 1
    for j in range(3):
        last = last * ((beta * sz[j] + gamma + wz[j]) % R_MOD) % R_MOD
    last = last * beta % R_MOD * P.grand_product_at_z_omega % R_MOD * alpha % R_MOD
    t = add(mul(P.grand_product_commitment, gz), neg(mul(vk.permutation_commitments[3], last)))
    d = mul(add(d, t), v)
    d = add(d, mul(P.grand_product_commitment, gzw))

    agg = P.quotient_poly_commitments[0]
    tf = 1
    for k in range(1, 4):
        tf = tf * zN % R_MOD
        agg = add(agg, mul(P.quotient_poly_commitments[k], tf))
    ch = v
    agg = add(agg, d)
    for c in P.wire_commitments:
        ch = ch * v % R_MOD
        agg = add(agg, mul(c, ch))
    for c in vk.permutation_commitments[:3]:
        ch = ch * v % R_MOD
        agg = add(agg, mul(c, ch))
    ch = ch * v % R_MOD
    ch = ch * v % R_MOD
    agg = add(agg, mul(P.wire_commitments[3], ch * u % R_MOD))
    ch = v
    val = (P.quotient_polynomial_at_z + P.linearization_polynomial_at_z * ch) % R_MOD
    for x in wz:
        ch = ch * v % R_MOD
        val = (val + x * ch) % R_MOD
    for x in sz:
        ch = ch * v % R_MOD
        val = (val + x * ch) % R_MOD
    ch = ch * v % R_MOD
    val = (val + P.grand_product_at_z_omega * ch % R_MOD * u) % R_MOD
    ch = ch * v % R_MOD
    val = (val + P.wire_values_at_z_omega[0] * ch % R_MOD * u) % R_MOD
    G = ol.g1_generator()
    agg = add(agg, neg(mul(G, val)))
    pg = add(agg, mul(P.opening_at_z_proof, z))
    pg = add(pg, mul(P.opening_at_z_omega_proof, z * om % R_MOD * u % R_MOD))
    px = neg(add(mul(P.opening_at_z_omega_proof, u), P.opening_at_z_proof))
    return ol.g1_is_inf(add(pg, mul(px, tau)))


# ===================================================================== synthetic circuits
class Xoshiro256ss:
    """xoshiro256** seeded through splitmix64 — the generator SURVEY.md §8(d) names for synthetic R1CS."""
    M = (1 << 64) - 1

    def __init__(self, seed):
        s, self.s = seed & self.M, []
        for _ in range(4):
            s = (s + 0x9E3779B97F4A7C15) & self.M
            zz = s
            zz = ((zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9) & self.M
            zz = ((zz ^ (zz >> 27)) * 0x94D049BB133111EB) & self.M
            self.s.append(zz ^ (zz >> 31))

    def next(self):
        s = self.s
        rot = lambda x, k: ((x << k) | (x >> (64 - k))) & self.M
        res = (rot((s[1] * 5) & self.M, 7) * 9) & self.M
        t = (s[1] << 17) & self.M
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = rot(s[3], 45)
        return res

    def fr(self):
        while True:
            v = (self.next() | (self.next() << 64) | (self.next() << 128) | (self.next() << 192)) & ((1 << 254) - 1)
            if v < R_MOD:
                return v
