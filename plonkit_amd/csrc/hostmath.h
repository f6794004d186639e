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
    F operator*(const F &o) const {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)l[j] * o.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * PR::INV;
            c = ((u128)m * PR::P[0] + t[0]) >> 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * PR::P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        if (t[4] || geq_p(t)) sub_p(t);
        F r; memcpy(r.l, t, 32); return r;
    }
    F sqr() const { return *this * *this; }
    F dbl() const { return *this + *this; }
    F pow(const uint64_t e[4]) const {
        F acc = one(), b = *this;
        for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * b; b = b.sqr(); }
        return acc;
    }
    F pow_u64(uint64_t e) const { uint64_t ee[4] = {e, 0, 0, 0}; return pow(ee); }
    F inv() const { uint64_t e[4] = {PR::P[0] - 2, PR::P[1], PR::P[2], PR::P[3]}; return pow(e); }
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
