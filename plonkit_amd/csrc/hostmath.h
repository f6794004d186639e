// Host-side (CPU) BN254 arithmetic used around the device kernels: final combination of MSM window
// sums, point (de)serialisation, transcript scalars, small per-round scalar computations.
// 4 x 64-bit limbs with unsigned __int128 — the same Montgomery representation (R = 2^256) as the
// device code and as ff_ce's Fr/Fq, so values move between the two by memcpy.
// This is product code (host half of the library), not the oracle: it never computes an NTT or MSM.
#pragma once
#include <stdint.h>
#include <string.h>

namespace plk {
namespace host {

typedef unsigned __int128 u128;

template <class PR>
struct F {
    uint64_t l[4];
    static F zero() { F r; memset(r.l, 0, 32); return r; }
    static F one() { F r; memcpy(r.l, PR::R, 32); return r; }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const F &o) const { return memcmp(l, o.l, 32) == 0; }
    bool operator!=(const F &o) const { return !(*this == o); }

    static bool geq_p(const uint64_t *t) {
        for (int i = 3; i >= 0; i--) { if (t[i] > PR::P[i]) return true; if (t[i] < PR::P[i]) return false; }
        return true;
    }
    static void sub_p(uint64_t *t) {
        uint64_t b = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - PR::P[i] - b; t[i] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1; }
    }
    F operator+(const F &o) const {
        F r; u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        if (geq_p(r.l)) sub_p(r.l);
        return r;
    }
    F operator-(const F &o) const {
        F r; uint64_t b = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)l[i] - o.l[i] - b; r.l[i] = (uint64_t)d; b = (uint64_t)(d >> 64) & 1; }
        if (b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + PR::P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    F operator-() const { return is_zero() ? *this : (zero() - *this); }
    // Montgomery product, operand scanning with the reduction interleaved limb by limb and fully unrolled.  Both BN254 moduli leave the top
    // two bits of their top limb clear, so the running value stays below 2p in four limbs: no fifth limb, no carry chain between the
    // multiplication and the reduction rows (30 ns against the 40 ns of the looped form with its six-limb accumulator; the host group
    // operations between two rounds of a proof are a few hundred of these).
    F operator*(const F &o) const {
        static_assert((PR::P[3] >> 62) == 0, "the no-carry form needs the two top bits of the modulus clear");
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        for (int i = 0; i < 4; i++) {
            const uint64_t y = o.l[i];
            u128 c = (u128)l[0] * y + t0;
            uint64_t a = (uint64_t)(c >> 64);
            const uint64_t m = (uint64_t)c * PR::INV;
            u128 d = (u128)m * PR::P[0] + (uint64_t)c;
            uint64_t r = (uint64_t)(d >> 64);
            c = (u128)l[1] * y + t1 + a; a = (uint64_t)(c >> 64); d = (u128)m * PR::P[1] + (uint64_t)c + r; t0 = (uint64_t)d; r = (uint64_t)(d >> 64);
            c = (u128)l[2] * y + t2 + a; a = (uint64_t)(c >> 64); d = (u128)m * PR::P[2] + (uint64_t)c + r; t1 = (uint64_t)d; r = (uint64_t)(d >> 64);
            c = (u128)l[3] * y + t3 + a; a = (uint64_t)(c >> 64); d = (u128)m * PR::P[3] + (uint64_t)c + r; t2 = (uint64_t)d; r = (uint64_t)(d >> 64);
            t3 = r + a;
        }
        F res; res.l[0] = t0; res.l[1] = t1; res.l[2] = t2; res.l[3] = t3;
        if (geq_p(res.l)) sub_p(res.l);
        return res;
    }
    F sqr() const { return *this * *this; }
    F dbl() const { return *this + *this; }
    F pow(const uint64_t e[4]) const {
        int top = 255;                                            // (x^N with N = 2^20 is 20 squarings, not 256)
        while (top >= 0 && !((e[top >> 6] >> (top & 63)) & 1)) top--;
        F acc = one(), b = *this;
        for (int i = 0; i <= top; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * b; if (i < top) b = b.sqr(); }
        return acc;
    }
    F pow_u64(uint64_t e) const { uint64_t ee[4] = {e, 0, 0, 0}; return pow(ee); }
    F inv_fermat() const { uint64_t e[4] = {PR::P[0] - 2, PR::P[1], PR::P[2], PR::P[3]}; return pow(e); }
    // Inverse.  Round 6: Bernstein-Yang division steps in batches of 62 on the low words (the variable-time "safegcd" scheme: per batch a 2x2
    // transition matrix from 62 cheap 64-bit steps, then one application of it to the five 62-bit limbs of f, g and of the Bezout pair d, e taken
    // modulo p) — ~2 us against the 13 us of the bit-by-bit binary Euclid it replaces (inv_euclid below); six host inversions sit between the
    // rounds of a proof, where the GPU waits for the next challenge (4 % of a 2^12-domain proof).  The result is VERIFIED with one product and the
    // old routine answers if it is ever wrong, so a slip in this code can cost time, never a proof.  inv(0) = 0, as before.
    F inv() const {
        if (is_zero()) return *this;
        static const F R3 = [] { F r2; memcpy(r2.l, PR::R2, 32); return r2 * r2; }();
        F r;
        if (inv_divsteps(l, r.l)) {
            r = r * R3;                                           // (aR)^-1 * R^3 / R = a^-1 R
            if ((*this * r) == one()) return r;
        }
        return inv_euclid();
    }
    typedef __int128 i128;
    // raw^-1 mod p for 0 < raw < p as plain integers; false if the 12 batches did not reach g = 0 (cannot happen for gcd(raw, p) = 1: 744 steps > the 735 bound)
    static bool inv_divsteps(const uint64_t raw[4], uint64_t out[4]) {
        const int64_t M62 = (int64_t)(~0ULL >> 2);
        auto to62 = [&](const uint64_t a[4], int64_t v[5]) {
            v[0] = (int64_t)(a[0] & (uint64_t)M62);
            v[1] = (int64_t)(((a[0] >> 62) | (a[1] << 2)) & (uint64_t)M62);
            v[2] = (int64_t)(((a[1] >> 60) | (a[2] << 4)) & (uint64_t)M62);
            v[3] = (int64_t)(((a[2] >> 58) | (a[3] << 6)) & (uint64_t)M62);
            v[4] = (int64_t)(a[3] >> 56);
        };
        int64_t P62[5], f[5], g[5], d[5] = {0, 0, 0, 0, 0}, e[5] = {1, 0, 0, 0, 0};
        to62(PR::P, P62); to62(PR::P, f); to62(raw, g);
        // p^-1 mod 2^62 (Newton: five doublings of precision from the 3 correct bits of p itself for odd p)
        uint64_t pinv = PR::P[0];
        for (int k = 0; k < 5; k++) pinv *= 2 - PR::P[0] * pinv;
        pinv &= (uint64_t)M62;
        int64_t eta = -1;
        for (int batch = 0; batch < 12; batch++) {
            // 62 division steps on the low words: [u v; q r] with 2^62 * [f', g'] = [u v; q r] [f, g]
            uint64_t u = 1, v = 0, q = 0, r = 1, ff = (uint64_t)f[0] | ((uint64_t)f[1] << 62), gg = (uint64_t)g[0] | ((uint64_t)g[1] << 62);
            int i = 62;
            for (;;) {
                const int zeros = __builtin_ctzll(gg | (~0ULL << i));
                gg >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
                if (i == 0) break;
                uint64_t m, w;
                int limit;
                if (eta < 0) {
                    uint64_t t;
                    eta = -eta;
                    t = ff; ff = gg; gg = 0 - t;
                    t = u; u = q; q = 0 - t;
                    t = v; v = r; r = 0 - t;
                    limit = (int)eta + 1 > i ? i : (int)eta + 1;
                    m = (~0ULL >> (64 - limit)) & 63u;
                    w = (ff * gg * (ff * ff - 2)) & m;
                } else {
                    limit = (int)eta + 1 > i ? i : (int)eta + 1;
                    m = (~0ULL >> (64 - limit)) & 15u;
                    w = ff + (((ff + 1) & 4) << 1);
                    w = (0 - w * gg) & m;
                }
                gg += ff * w; q += u * w; r += v * w;
            }
            const int64_t tu = (int64_t)u, tv = (int64_t)v, tq = (int64_t)q, tr = (int64_t)r;
            {   // [d, e] <- [u v; q r] [d, e] / 2^62 (mod p): multiples of p make the low 62 bits vanish first
                const int64_t sd = d[4] >> 63, se = e[4] >> 63;
                int64_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
                i128 cd = (i128)tu * d[0] + (i128)tv * e[0], ce = (i128)tq * d[0] + (i128)tr * e[0];
                md -= (int64_t)((pinv * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);
                me -= (int64_t)((pinv * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
                cd += (i128)P62[0] * md; ce += (i128)P62[0] * me;
                cd >>= 62; ce >>= 62;
                for (int k = 1; k < 5; k++) {
                    cd += (i128)tu * d[k] + (i128)tv * e[k] + (i128)P62[k] * md;
                    ce += (i128)tq * d[k] + (i128)tr * e[k] + (i128)P62[k] * me;
                    d[k - 1] = (int64_t)cd & M62; cd >>= 62;
                    e[k - 1] = (int64_t)ce & M62; ce >>= 62;
                }
                d[4] = (int64_t)cd; e[4] = (int64_t)ce;
            }
            {   // [f, g] <- [u v; q r] [f, g] / 2^62 (exact)
                i128 cf = (i128)tu * f[0] + (i128)tv * g[0], cg = (i128)tq * f[0] + (i128)tr * g[0];
                cf >>= 62; cg >>= 62;
                for (int k = 1; k < 5; k++) {
                    cf += (i128)tu * f[k] + (i128)tv * g[k];
                    cg += (i128)tq * f[k] + (i128)tr * g[k];
                    f[k - 1] = (int64_t)cf & M62; cf >>= 62;
                    g[k - 1] = (int64_t)cg & M62; cg >>= 62;
                }
                f[4] = (int64_t)cf; g[4] = (int64_t)cg;
            }
            if ((g[0] | g[1] | g[2] | g[3] | g[4]) == 0) {
                // f = +-1 = gcd (limbs {1,0,0,0,0} or {M62,M62,M62,M62,-1}); the inverse is d * f, brought from (-2p, p) into [0, p)
                auto is_neg = [&](const int64_t a[5]) { return a[4] < 0; };
                auto addp = [&](int64_t a[5], int sign) {          // a += sign * p, carries normalised (the top limb keeps the sign)
                    int64_t c = 0;
                    for (int k = 0; k < 5; k++) { const int64_t t = a[k] + (sign > 0 ? P62[k] : -P62[k]) + c; if (k < 4) { a[k] = t & M62; c = t >> 62; } else a[k] = t; }
                };
                auto negate = [&](int64_t a[5]) { int64_t c = 0; for (int k = 0; k < 5; k++) { const int64_t t = -a[k] + c; if (k < 4) { a[k] = t & M62; c = t >> 62; } else a[k] = t; } };
                const bool minus = f[4] < 0;
                if (minus) negate(d);
                for (int guard = 0; guard < 4 && is_neg(d); guard++) addp(d, +1);
                for (int guard = 0; guard < 4; guard++) {          // d >= p ?  subtract
                    int64_t t[5]; for (int k = 0; k < 5; k++) t[k] = d[k];
                    addp(t, -1);
                    if (is_neg(t)) break;
                    for (int k = 0; k < 5; k++) d[k] = t[k];
                }
                if (is_neg(d)) return false;
                out[0] = (uint64_t)d[0] | ((uint64_t)d[1] << 62);
                out[1] = ((uint64_t)d[1] >> 2) | ((uint64_t)d[2] << 60);
                out[2] = ((uint64_t)d[2] >> 4) | ((uint64_t)d[3] << 58);
                out[3] = ((uint64_t)d[3] >> 6) | ((uint64_t)d[4] << 56);
                return true;
            }
        }
        return false;
    }
    // the binary extended Euclid (Hankerson-Menezes-Vanstone, Alg. 2.22) on the stored limbs, then one Montgomery product by R^3 to come back to
    // Montgomery form — the routine of rounds 4-5, now the checked fallback of inv()
    F inv_euclid() const {
        if (is_zero()) return *this;
        static const F R3 = [] { F r2; memcpy(r2.l, PR::R2, 32); return r2 * r2; }();
        auto is_one = [](const uint64_t *t) { return t[0] == 1 && (t[1] | t[2] | t[3]) == 0; };
        auto even = [](const uint64_t *t) { return (t[0] & 1) == 0; };
        auto shr1 = [](uint64_t *t) { t[0] = (t[0] >> 1) | (t[1] << 63); t[1] = (t[1] >> 1) | (t[2] << 63); t[2] = (t[2] >> 1) | (t[3] << 63); t[3] >>= 1; };
        auto add_p = [](uint64_t *t) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + PR::P[i]; t[i] = (uint64_t)c; c >>= 64; } };   // p < 2^254: no carry out
        auto geq = [](const uint64_t *a, const uint64_t *b) { for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return true; };
        auto sub = [](uint64_t *a, const uint64_t *b) { uint64_t br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; a[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return br; };
        uint64_t u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
        memcpy(u, l, 32); memcpy(v, PR::P, 32);
        while (!is_one(u) && !is_one(v)) {
            while (even(u)) { shr1(u); if (!even(x1)) add_p(x1); shr1(x1); }
            while (even(v)) { shr1(v); if (!even(x2)) add_p(x2); shr1(x2); }
            if (geq(u, v)) { sub(u, v); if (sub(x1, x2)) add_p(x1); }
            else { sub(v, u); if (sub(x2, x1)) add_p(x2); }
        }
        F r; memcpy(r.l, is_one(u) ? x1 : x2, 32);
        return r * R3;
    }
    static F from_canonical(const uint64_t c[4]) { F t, rr; memcpy(t.l, c, 32); memcpy(rr.l, PR::R2, 32); return t * rr; }
    static F from_u64(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; return from_canonical(c); }
    void to_canonical(uint64_t out[4]) const { F o = zero(); o.l[0] = 1; F r = *this * o; memcpy(out, r.l, 32); }
    // 32-byte big-endian canonical
    void to_be_bytes(uint8_t out[32]) const {
        uint64_t c[4]; to_canonical(c);
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(c[i] >> (8 * b));
    }
    // returns false when the value is >= p
    static bool from_be_bytes(const uint8_t in[32], F *out) {
        uint64_t c[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) c[i] |= (uint64_t)in[31 - (8 * i + b)] << (8 * b);
        if (geq_p(c)) return false;
        *out = from_canonical(c);
        return true;
    }
};

struct FrP {
    static constexpr uint64_t P[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R[4] = {0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;
};
struct FqP {
    static constexpr uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    static constexpr uint64_t R[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
    static constexpr uint64_t R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
    static constexpr uint64_t INV = 0x87d20782e4866389ULL;
};
typedef F<FrP> HFr;
typedef F<FqP> HFq;

struct HAffine { HFq x, y; bool is_inf() const { return x.is_zero() && y.is_zero(); } };
struct HJac {
    HFq x, y, z;
    static HJac inf() { HJac r; r.x = HFq::one(); r.y = HFq::one(); r.z = HFq::zero(); return r; }
    bool is_inf() const { return z.is_zero(); }
};

inline HJac jac_from_affine(const HAffine &a) {
    if (a.is_inf()) return HJac::inf();
    HJac r; r.x = a.x; r.y = a.y; r.z = HFq::one(); return r;
}
inline HJac jac_double(const HJac &p) {
    if (p.is_inf()) return p;
    HFq A = p.x.sqr(), B = p.y.sqr(), C = B.sqr();
    HFq D = ((p.x + B).sqr() - A - C).dbl();
    HFq E = A.dbl() + A, Fv = E.sqr();
    HJac r;
    r.x = Fv - D.dbl();
    r.z = (p.y * p.z).dbl();
    r.y = E * (D - r.x) - C.dbl().dbl().dbl();
    return r;
}
inline HJac jac_add(const HJac &p, const HJac &q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    HFq z1z1 = p.z.sqr(), z2z2 = q.z.sqr();
    HFq u1 = p.x * z2z2, u2 = q.x * z1z1;
    HFq s1 = p.y * q.z * z2z2, s2 = q.y * p.z * z1z1;
    if (u1 == u2) return (s1 == s2) ? jac_double(p) : HJac::inf();
    HFq h = u2 - u1, i = h.dbl().sqr(), j = h * i, rr = (s2 - s1).dbl(), v = u1 * i;
    HJac r;
    r.x = rr.sqr() - j - v.dbl();
    r.y = rr * (v - r.x) - (s1 * j).dbl();
    r.z = ((p.z + q.z).sqr() - z1z1 - z2z2) * h;
    return r;
}
inline HAffine jac_to_affine(const HJac &p) {
    HAffine a;
    if (p.is_inf()) { a.x = HFq::zero(); a.y = HFq::zero(); return a; }
    HFq zi = p.z.inv(), zi2 = zi.sqr();
    a.x = p.x * zi2; a.y = p.y * zi2 * zi;
    return a;
}
// n points with ONE inversion (Montgomery's trick on the z coordinates; points at infinity are skipped)
inline void jac_to_affine_batch(const HJac *p, uint32_t n, HAffine *out) {
    HFq prefix[16];
    if (n > 16) { for (uint32_t k = 0; k < n; k++) out[k] = jac_to_affine(p[k]); return; }
    HFq acc = HFq::one();
    for (uint32_t k = 0; k < n; k++) { prefix[k] = acc; if (!p[k].is_inf()) acc = acc * p[k].z; }
    HFq inv = acc.inv();
    for (uint32_t k = n; k-- > 0;) {
        if (p[k].is_inf()) { out[k].x = HFq::zero(); out[k].y = HFq::zero(); continue; }
        const HFq zi = inv * prefix[k], zi2 = zi.sqr();
        inv = inv * p[k].z;
        out[k].x = p[k].x * zi2; out[k].y = p[k].y * zi2 * zi;
    }
}
inline HJac jac_neg(const HJac &p) { HJac r = p; r.y = -p.y; return r; }
// k canonical little-endian limbs
inline HJac jac_mul(const HJac &p, const uint64_t k[4]) {
    HJac acc = HJac::inf();
    for (int i = 255; i >= 0; i--) { acc = jac_double(acc); if ((k[i >> 6] >> (i & 63)) & 1) acc = jac_add(acc, p); }
    return acc;
}
inline bool on_curve(const HAffine &a) {
    if (a.is_inf()) return true;
    return a.y.sqr() == a.x.sqr() * a.x + HFq::from_u64(3);
}

}  // namespace host
}  // namespace plk
