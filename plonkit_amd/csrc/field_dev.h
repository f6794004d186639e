// BN254 prime-field arithmetic for gfx950 (and the host side of the same library).
//
// Replaces what the reference reaches through ff_ce 0.12.0 / pairing_ce 0.24.2 `Fr`/`Fq`
// (Cargo.lock:594-596,1212-1214): 256-bit Montgomery arithmetic, R = 2^256.  In-memory layout is
// identical to ff_ce's (4 x u64 little-endian limbs, Montgomery form), so vectors cross the C ABI
// unchanged; on the device the same 32 bytes are read as 8 x u32 limbs because CDNA4's integer
// multiplier is 32 x 32 (v_mad_u64_u32 / v_mul_hi_u32) — there is no 64-bit multiply and no MFMA
// path for modular integer arithmetic.
//
// Both moduli are < 2^254, so sums of two reduced values never overflow 2^256 and the CIOS
// accumulator never needs a 9th limb.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PLK_HD __host__ __device__ __forceinline__

namespace plk {

struct FrParams {
    // r = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                      0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
    static constexpr uint32_t INV = 0xefffffffu;   // -r^-1 mod 2^32
};

struct FqParams {
    // q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t R[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                      0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    static constexpr uint32_t INV = 0xe4866389u;   // -q^-1 mod 2^32
};

template <class PR>
struct alignas(16) Fp {
    uint32_t l[8];

    static PLK_HD Fp zero() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
    static PLK_HD Fp one() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = PR::R[i]; return r; }
    static PLK_HD Fp r2() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = PR::R2[i]; return r; }

    PLK_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= l[i]; return o == 0; }
    PLK_HD bool operator==(const Fp &b) const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i]; return o == 0; }
    PLK_HD bool operator!=(const Fp &b) const { return !(*this == b); }
};

// t -= p if t >= p  (t < 2p on entry)
template <class PR>
PLK_HD void reduce_once(uint32_t t[8]) {
    uint32_t s[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)t[i] - PR::P[i] - br;
        s[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    if (!br) {
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = s[i];
    }
}

template <class PR>
PLK_HD Fp<PR> add(const Fp<PR> &a, const Fp<PR> &b) {
    Fp<PR> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
    reduce_once<PR>(r.l);
    return r;
}

template <class PR>
PLK_HD Fp<PR> sub(const Fp<PR> &a, const Fp<PR> &b) {
    Fp<PR> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.l[i] - b.l[i] - br; r.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)r.l[i] + (PR::P[i] & mask); r.l[i] = (uint32_t)c; c >>= 32; }
    return r;
}

template <class PR>
PLK_HD Fp<PR> neg(const Fp<PR> &a) {
    Fp<PR> r;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nz |= a.l[i];
    uint32_t mask = nz ? 0xffffffffu : 0u;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)(PR::P[i] & mask) - a.l[i] - br; r.l[i] = (uint32_t)d; br = (d >> 32) & 1; }
    return r;
}

template <class PR>
PLK_HD Fp<PR> dbl(const Fp<PR> &a) { return add(a, a); }

// Montgomery product a*b*R^-1 mod p, fully reduced.  CIOS over 32-bit limbs: every inner step is
// one 32x32+64 multiply-add (v_mad_u64_u32 on gfx950); 64 + 64 of them plus 8 low multiplies.
template <class PR>
PLK_HD Fp<PR> mul(const Fp<PR> &a, const Fp<PR> &b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t t8 = t[8] + (uint32_t)c;            // never carries: p < 2^254
        uint32_t m = t[0] * PR::INV;
        c = ((uint64_t)m * PR::P[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = (uint64_t)m * PR::P[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t8;
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    reduce_once<PR>(t);
    Fp<PR> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    return r;
}

template <class PR>
PLK_HD Fp<PR> sqr(const Fp<PR> &a) { return mul(a, a); }

template <class PR>
PLK_HD Fp<PR> from_canonical(const Fp<PR> &a) { return mul(a, Fp<PR>::r2()); }

template <class PR>
PLK_HD Fp<PR> to_canonical(const Fp<PR> &a) {
    Fp<PR> o = Fp<PR>::zero();
    o.l[0] = 1;
    return mul(a, o);
}

template <class PR>
PLK_HD Fp<PR> from_u64(uint64_t v) {
    Fp<PR> o = Fp<PR>::zero();
    o.l[0] = (uint32_t)v;
    o.l[1] = (uint32_t)(v >> 32);
    return from_canonical(o);
}

// a^e for a 64-bit exponent (not constant time; exponents here are public domain indices)
template <class PR>
PLK_HD Fp<PR> pow_u64(Fp<PR> a, uint64_t e) {
    Fp<PR> acc = Fp<PR>::one();
    while (e) {
        if (e & 1) acc = mul(acc, a);
        a = sqr(a);
        e >>= 1;
    }
    return acc;
}

// Fermat inversion a^(p-2); inv(0) = 0
template <class PR>
PLK_HD Fp<PR> inv(const Fp<PR> &a) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = PR::P[i];
    e[0] -= 2;                                        // low limb of both moduli is >= 2
    Fp<PR> acc = Fp<PR>::one(), base = a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    return acc;
}

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// 16-byte vector view used for coalesced loads/stores of field elements (2 x dwordx4 per element)
struct alignas(16) u32x4 { uint32_t x, y, z, w; };

template <class PR>
__device__ __forceinline__ Fp<PR> load_fp(const Fp<PR> *p) {
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
    u32x4 lo = q[0], hi = q[1];
    Fp<PR> r;
    r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
    r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
    return r;
}

template <class PR>
__device__ __forceinline__ void store_fp(Fp<PR> *p, const Fp<PR> &v) {
    u32x4 *q = reinterpret_cast<u32x4 *>(p);
    q[0] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
    q[1] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
}

}  // namespace plk
