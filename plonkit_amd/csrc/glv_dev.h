// GLV decomposition of a BN254 Fr scalar for the one place in this library that multiplies a VARIABLE base by a full-width
// scalar: the butterflies of the inverse NTT over G1 (g1ntt.hip — Crs::<Lagrange>::from_powers, src/plonk.rs:179-185).
// (The commitments of the prover use a fixed-base table: all their doublings are precomputed and GLV has nothing to remove.)
//
// BN254's G1 (y^2 = x^3 + 3 over Fq, prime order r) carries the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) with
// beta^3 = 1 in Fq and lambda^3 = 1 in Fr.  Every k < r splits as k = k1 + k2 lambda (mod r) with |k1|, |k2| < 2^128, so
// k P = k1 P + k2 phi(P) costs one shared chain of 129 doublings instead of 255.
//
// Constants (derived with Python integers, checked on the host by tests/test_field29_host.py and on the GPU by the G1 iNTT
// tests against the oracle):
//   lambda = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
//   beta   = 0x30644e72e131a0295e6dd9e7e0acccb0c28f069fbb966e3de4bd44e5607cfd48
//   lattice {(a, b): a + b lambda = 0 mod r}, reduced basis (a1, b1) = (A1, -B1N), (a2, b2) = (A2, B2), det = r:
//     A1 = 0x6f4d8248eeb859fc8211bbeb7d4f1128   B1N = 0x89d3256894d213e3
//     A2 = 0x89d3256894d213e3                   B2  = 0x6f4d8248eeb859fd0be4e1541221250b
//   c1 = floor(k * G1 / 2^256), G1 = floor(2^256 B2 / r);  c2 = floor(k * G2 / 2^256), G2 = floor(2^256 B1N / r)
//   k1 = k - c1 A1 - c2 A2  in [0, A1 + A2) ;  k2 = c1 B1N - c2 B2  in (-B1N, B2)      (floor instead of round: one basis
//   vector more at worst — both halves stay below 2^128, which is what the 43 signed 3-bit windows of the caller hold)
#pragma once
#include "field_dev.h"

namespace plk {

struct GlvSplit {
    uint32_t k1[5], k2[5];       // magnitudes, < 2^128 (the fifth limb is zero; kept for the window reader)
    bool neg1, neg2;
};

namespace glv {
constexpr uint32_t G1[5] = {0x00ff6565u, 0x5398fd03u, 0xa773d2d2u, 0x4ccef014u, 0x2u};
constexpr uint32_t G2[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x2u};
constexpr uint32_t A1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};
constexpr uint32_t A2[2] = {0x94d213e3u, 0x89d32568u};
constexpr uint32_t B1N[2] = {0x94d213e3u, 0x89d32568u};
constexpr uint32_t B2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
// beta in the 2^261 Montgomery domain of the 29-bit layer (field29_dev.h), 9 x 29-bit limbs
constexpr uint32_t BETA_W[9] = {0x18ccb791u, 0x175b1c3au, 0x0b83d6e2u, 0x0e8ed071u, 0x1282bee2u, 0x04220e84u, 0x1fe4017fu, 0x15084d4au, 0x00169119u};

// limbs [SKIP, SKIP + NO) of a * b (a: NA limbs, b: NB limbs), schoolbook with a 64-bit column accumulator + carry word
template <int NA, int NB, int SKIP, int NO>
PLK_HD void mul_window(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint64_t acc = 0;            // column sum (low 64 bits) ...
    uint32_t hi = 0;             // ... and its overflow
#pragma unroll
    for (int col = 0; col < SKIP + NO; col++) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int j = col - i;
            if (j < 0 || j >= NB) continue;
            const uint64_t p = (uint64_t)a[i] * b[j];
            acc += p;
            hi += acc < p ? 1u : 0u;
        }
        if (col >= SKIP) out[col - SKIP] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
}
// out = a - b (N limbs, wrap-around)
template <int N> PLK_HD void sub_n(uint32_t *out, const uint32_t *a, const uint32_t *b) {
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { const uint64_t d = (uint64_t)a[i] - b[i] - br; out[i] = (uint32_t)d; br = (d >> 32) & 1u; }
}
template <int N> PLK_HD void add_n(uint32_t *out, const uint32_t *a, const uint32_t *b) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { const uint64_t s = (uint64_t)a[i] + b[i] + c; out[i] = (uint32_t)s; c = s >> 32; }
}
template <int N> PLK_HD bool abs_n(uint32_t *v) {               // two's complement -> (sign, magnitude)
    const bool negative = (v[N - 1] >> 31) != 0;
    if (negative) {
        uint64_t c = 1;
#pragma unroll
        for (int i = 0; i < N; i++) { const uint64_t s = (uint64_t)(~v[i]) + c; v[i] = (uint32_t)s; c = s >> 32; }
    }
    return negative;
}
}  // namespace glv

// k: CANONICAL (non-Montgomery) scalar < r as 8 x 32-bit limbs
PLK_HD GlvSplit glv_split(const uint32_t k[8]) {
    using namespace glv;
    uint32_t c1[4], c2[3];
    mul_window<8, 5, 8, 4>(k, G1, c1);                            // c1 < 2^127
    mul_window<8, 3, 8, 3>(k, G2, c2);                            // c2 < 2^64 (third limb zero)
    // everything below is exact modulo 2^160; the true values lie in (-2^128, 2^129)
    uint32_t t1[5], t2[5], klo[5], s[5];
    GlvSplit r;
    mul_window<4, 4, 0, 5>(c1, A1, t1);
    mul_window<3, 2, 0, 5>(c2, A2, t2);
#pragma unroll
    for (int i = 0; i < 5; i++) klo[i] = k[i];
    sub_n<5>(s, klo, t1);
    sub_n<5>(r.k1, s, t2);
    r.neg1 = abs_n<5>(r.k1);
    mul_window<4, 2, 0, 5>(c1, B1N, t1);
    mul_window<3, 4, 0, 5>(c2, B2, t2);
    sub_n<5>(r.k2, t1, t2);
    r.neg2 = abs_n<5>(r.k2);
    return r;
}

// signed 3-bit windows of a magnitude < 2^128 (five limbs), low to high: digits in [-3, 4] as 4-bit codes (bit 3 = negative,
// bits 0-2 = magnitude), 43 of them packed eight to a word.  v = window + carry; v <= 4 -> digit v; v >= 5 -> v - 8, carry 1
// (the top window holds bits 126..128 <= 3, so no carry leaves it).
PLK_HD void glv_digits(const uint32_t k[5], uint32_t dig[6]) {
#pragma unroll
    for (int i = 0; i < 6; i++) dig[i] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < 43; w++) {
        const int pos = 3 * w, limb = pos >> 5, off = pos & 31;
        uint64_t two = k[limb];
        if (limb + 1 < 5) two |= (uint64_t)k[limb + 1] << 32;
        const uint32_t v = ((uint32_t)(two >> off) & 7u) + carry;
        uint32_t code;
        if (v >= 5) { code = 8u | (8u - v); carry = 1; } else { code = v; carry = 0; }
        dig[w >> 3] |= code << (4 * (w & 7));
    }
}

// signed 4-bit windows of a magnitude < 2^127 (both GLV halves are: k1 < A1 + A2, |k2| < max(B1N, B2), all below 2^127), low to high: digits in [-7, 8] as
// 5-bit codes (bit 4 = negative, bits 0-3 = magnitude), 32 of them packed six to a word.  v = window + carry; v <= 8 -> digit v; v >= 9 -> v - 16, carry 1
// (the top window holds bits 124..126 <= 7, so no carry leaves it).
PLK_HD void glv_digits4(const uint32_t k[5], uint32_t dig[6]) {
#pragma unroll
    for (int i = 0; i < 6; i++) dig[i] = 0;
    uint32_t carry = 0;
#pragma unroll
    for (int w = 0; w < 32; w++) {
        const uint32_t v = ((k[w >> 3] >> (4 * (w & 7))) & 15u) + carry;
        uint32_t code;
        if (v >= 9) { code = 16u | (16u - v); carry = 1; } else { code = v; carry = 0; }
        dig[w / 6] |= code << (5 * (w % 6));
    }
}

}  // namespace plk
