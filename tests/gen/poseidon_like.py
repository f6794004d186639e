"""Poseidon-SHAPED synthetic R1CS for the tests of the long-linear-combination transpiler paths.

The reference's own poseidon artifacts (test/test_poseidon_plonk.sh builds them with circom / snarkjs) are not in its
tree and cannot be generated here, so this generator reproduces what circom emits for a Poseidon hash chain — the SHAPES,
not circomlib's constants: width-3 state, RF full rounds and RP partial rounds of x^5 S-boxes
    (lc) * (lc) = x2 ;  x2 * x2 = x4 ;  x4 * (lc) = x5
whose inputs are LINEAR COMBINATIONS of earlier signals (the MDS mix and the round constants are substituted into the
next S-box by circom's optimiser, so the un-S-boxed lanes of the partial rounds grow to dozens of terms), and a final
    (lc) * 1 = out
per output lane (constant x LC = LC).  MDS entries and round constants come from xoshiro256**; the witness is computed
alongside.  Everything the transpilers do with such constraints (d / d_next chains, constant merges) is UNPINNED: no
reference fixture reaches it — tests built on this check the product against the oracle restatement and
against the verifier, never against the reference."""
# (self-contained on purpose: bench.py's prove.by_domain leg builds its 2^12 circuit with this generator, and nothing on a GPU-timed path may import
#  oracle/ — the generator only makes inputs.  Same xoshiro256** / splitmix64 as oracle.plonk_oracle.Xoshiro256ss: tests/test_host_abi.py checks that.)
R_MOD = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001          # BN254 Fr


class Xoshiro256ss:
    """xoshiro256** seeded through splitmix64 — the generator SURVEY.md §8(d) names for synthetic R1CS"""
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


class _Lc:
    """linear combination: {wire: coeff} + constant (wire 0 is ONE)"""

    def __init__(self, terms=None, const=0):
        self.t, self.c = dict(terms or {}), const % R_MOD

    def scaled(self, k):
        return _Lc({w: c * k % R_MOD for w, c in self.t.items()}, self.c * k)

    def plus(self, o):
        t = dict(self.t)
        for w, c in o.t.items():
            t[w] = (t.get(w, 0) + c) % R_MOD
        return _Lc({w: c for w, c in t.items() if c}, self.c + o.c)

    def value(self, wit):
        return (self.c + sum(c * wit[w] for w, c in self.t.items())) % R_MOD

    def as_list(self):
        out = [(w, c) for w, c in sorted(self.t.items())]
        if self.c:
            out = [(0, self.c)] + out
        return out


def build(n_perms, seed, rf=8, rp=20):
    """returns (num_inputs, num_variables, constraints, witness): one public input (the last hash output), two private
    inputs, a chain of n_perms permutations (the capacity lane carries over)"""
    rng = Xoshiro256ss(seed)
    wit = [1, 0, rng.fr(), rng.fr()]                # ONE, public output (patched at the end), two private inputs
    cons = []

    def new_signal(v):
        wit.append(v % R_MOD)
        return len(wit) - 1

    def sbox(lc):
        v = lc.value(wit)
        x2 = new_signal(v * v)
        cons.append((lc.as_list(), lc.as_list(), [(x2, 1)]))
        x4 = new_signal(wit[x2] * wit[x2])
        cons.append(([(x2, 1)], [(x2, 1)], [(x4, 1)]))
        x5 = new_signal(wit[x4] * v)
        cons.append(([(x4, 1)], lc.as_list(), [(x5, 1)]))
        return _Lc({x5: 1})

    mds = [[rng.fr() for _ in range(3)] for _ in range(3)]
    state = [_Lc({2: 1}), _Lc({3: 1}), _Lc({}, 0)]
    for _ in range(n_perms):
        for rnd in range(rf + rp):
            state = [s.plus(_Lc({}, rng.fr())) for s in state]                      # add round constants
            full = rnd < rf // 2 or rnd >= rf // 2 + rp
            state = [sbox(s) if (full or i == 0) else s for i, s in enumerate(state)]
            mixed = []
            for i in range(3):
                acc = _Lc()
                for j in range(3):
                    acc = acc.plus(state[j].scaled(mds[i][j]))
                mixed.append(acc)
            state = mixed
        outs = []
        for s in state:                                                             # (lc) * 1 = out : constant x LC
            o = new_signal(s.value(wit))
            cons.append((s.as_list(), [(0, 1)], [(o, 1)]))
            outs.append(_Lc({o: 1}))
        state = outs
    last = next(iter(state[0].t))
    wit[1] = wit[last]
    cons.append(([(1, 1)], [(0, 1)], [(last, 1)]))                                  # ties the public input to the digest
    return 2, len(wit), cons, wit


def as_circom_json(num_inputs, num_vars, cons):
    return {"n8": 32, "prime": str(R_MOD), "nVars": num_vars, "nOutputs": 0, "nPubInputs": num_inputs - 1, "nPrvInputs": 2,
            "nLabels": num_vars, "nConstraints": len(cons),
            "constraints": [[{str(w): str(c) for w, c in lc} for lc in con] for con in cons]}
