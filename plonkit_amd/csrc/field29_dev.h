// Carry-free 9 x 29-bit-limb arithmetic for the VALU-bound kernels (NTT butterflies, MSM).
//
// Why: on gfx950 every carry-producing VALU op (v_add_co/v_addc, 64-bit adds) costs the same
// ~4.2 cycles per wave as a v_mad_u64_u32, while plain v_add_u32/v_and/v_lshr cost ~2.25
// (profiles/r01_ubench_int2.txt).  The 8 x 32-bit CIOS product of field_dev.h spends 128 mads + 130
// 64-bit adds + 275 moves per product (~1600 cycles per wave).  With 29-bit limbs a column of
// 18 partial products fits a 64-bit accumulator without any carry handling: 162 mads + a short
// normalisation (~900 cycles), and additions/subtractions become 9 plain 32-bit ops.
//
// Representation W: value = sum l[i] * 2^(29 i), 9 limbs, capacity 2^261 = ~170 p.
// Montgomery radix inside this layer is R' = 2^261 (nine 29-bit reduction steps):
//     mulw(a, b) = a * b * 2^-261 mod p,   result < a*b/2^261 + p   (not fully reduced)
// p / 2^261 = 0.0059, so sums of several un-reduced values can be multiplied again without any
// conditional subtraction ("lazy reduction"); bounds are tracked per call site.
// Data in HBM keeps the library's external form (8 x u32, R = 2^256): the NTT is linear, so it can
// run W arithmetic on the raw 256-bit values as long as the twiddle CONSTANTS are in the 2^261
// domain; the MSM keeps its resident SRS and its accumulators in the 2^261 domain and converts
// at the boundary.  No MFMA: these are 29x29->58-bit integer multiply-adds on the VALU.
#pragma once
#include "field_dev.h"
#include <utility>

namespace plk {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}) — `#pragma unroll` is only a
// request, and a loop the compiler leaves rolled puts the register arrays it indexes into scratch memory
template <int... J, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, J...>, F &&f) { (f(std::integral_constant<int, J>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f)); }

constexpr uint32_t M29 = (1u << 29) - 1;
constexpr uint32_t MULW_A_LIMB_MAX = 3280000000u;          // largest limb of mulw's LEFT operand (derivation at mulw)

struct FrW {
    static constexpr uint32_t P29[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV29 = 0x0fffffffu;                   // -p^-1 mod 2^29
    static constexpr uint32_t PINV0 = 0x10000001u;                   //  p^-1 mod 2^29
    static constexpr uint32_t ONE_W[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t W_FROM_S[9] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    static constexpr uint32_t S_FROM_W[9] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    static constexpr uint32_t PAD2[9] = {0x80000002u, 0x9e1f593bu, 0x9cb8489du, 0x8fa121e2u, 0x8b0ba502u, 0x85b6817du, 0x814dc27eu, 0x9cb84c64u, 0x0060c898u};
    static constexpr uint32_t PAD4[9] = {0x80000004u, 0x9c3eb27au, 0x9970913fu, 0x9f4243c9u, 0x96174a08u, 0x8b6d02feu, 0x829b8500u, 0x997098ccu, 0x00c19135u};
    static constexpr uint32_t PAD6[9] = {0x80000006u, 0x9a5e0bb9u, 0x9628d9e1u, 0x8ee365b0u, 0x8122ef0fu, 0x91238480u, 0x83e94782u, 0x9628e534u, 0x012259d2u};
    static constexpr uint32_t PAD8[9] = {0x80000008u, 0x987d64f8u, 0x92e12283u, 0x9e848797u, 0x8c2e9415u, 0x96da0601u, 0x85370a04u, 0x92e1319cu, 0x0183226fu};
};

struct FqW {
    static constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    static constexpr uint32_t INV29 = 0x04866389u;
    static constexpr uint32_t PINV0 = 0x1b799c77u;
    static constexpr uint32_t ONE_W[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    static constexpr uint32_t W_FROM_S[9] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
    static constexpr uint32_t S_FROM_W[9] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    static constexpr uint32_t PAD2[9] = {0x90f9fa8eu, 0x8208c169u, 0x98e5469au, 0x85aa459du, 0x8b0bb2ecu, 0x85b6817du, 0x814dc27eu, 0x9cb84c64u, 0x0060c898u};
    static constexpr uint32_t PAD4[9] = {0x81f3f51cu, 0x841182d7u, 0x91ca8d38u, 0x8b548b3fu, 0x961765dcu, 0x8b6d02feu, 0x829b8500u, 0x997098ccu, 0x00c19135u};
    static constexpr uint32_t PAD6[9] = {0x92edefaau, 0x861a4444u, 0x8aafd3d6u, 0x90fed0e1u, 0x812318ccu, 0x91238480u, 0x83e94782u, 0x9628e534u, 0x012259d2u};
    static constexpr uint32_t PAD8[9] = {0x83e7ea38u, 0x882305b2u, 0x83951a74u, 0x96a91683u, 0x8c2ecbbcu, 0x96da0601u, 0x85370a04u, 0x92e1319cu, 0x0183226fu};
};

template <class WP>
struct W9 {
    uint32_t l[9];
};

template <class WP> PLK_HD W9<WP> w_zero() { W9<WP> r; for (int i = 0; i < 9; i++) r.l[i] = 0; return r; }
template <class WP> PLK_HD W9<WP> w_one() { W9<WP> r; for (int i = 0; i < 9; i++) r.l[i] = WP::ONE_W[i]; return r; }
template <class WP> PLK_HD W9<WP> w_from_s_const() { W9<WP> r; for (int i = 0; i < 9; i++) r.l[i] = WP::W_FROM_S[i]; return r; }
template <class WP> PLK_HD W9<WP> s_from_w_const() { W9<WP> r; for (int i = 0; i < 9; i++) r.l[i] = WP::S_FROM_W[i]; return r; }
template <class WP> PLK_HD bool w_all_zero(const W9<WP> &a) { uint32_t o = 0; for (int i = 0; i < 9; i++) o |= a.l[i]; return o == 0; }

// 8 x u32 packed -> 9 x 29 (no arithmetic; the 256-bit integer is re-sliced)
template <class WP, class PR>
PLK_HD W9<WP> unpack(const Fp<PR> &a) {
    W9<WP> r;
    const uint32_t *w = a.l;
    r.l[0] = w[0] & M29;
    r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
    r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
    r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
    r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
    r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
    r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
    r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
    r.l[8] = w[7] >> 8;
    return r;
}

// normalised limbs, value < 2^256  ->  8 x u32
template <class PR, class WP>
PLK_HD Fp<PR> pack(const W9<WP> &a) {
    Fp<PR> r;
    const uint32_t *l = a.l;
    r.l[0] = l[0] | (l[1] << 29);
    r.l[1] = (l[1] >> 3) | (l[2] << 26);
    r.l[2] = (l[2] >> 6) | (l[3] << 23);
    r.l[3] = (l[3] >> 9) | (l[4] << 20);
    r.l[4] = (l[4] >> 12) | (l[5] << 17);
    r.l[5] = (l[5] >> 15) | (l[6] << 14);
    r.l[6] = (l[6] >> 18) | (l[7] << 11);
    r.l[7] = (l[7] >> 21) | (l[8] << 8);
    return r;
}

// carry propagation: limbs (any u32, total value < 2^261) -> limbs < 2^29
template <class WP>
PLK_HD W9<WP> normw(const W9<WP> &a) {
    W9<WP> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint32_t s = a.l[i] + c; r.l[i] = s & M29; c = s >> 29; }
    r.l[8] = a.l[8] + c;
    return r;
}

// limb-wise sum, no carry handling (limbs grow by one bit)
template <class WP>
PLK_HD W9<WP> addw(const W9<WP> &a, const W9<WP> &b) {
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + k*p, where k*p is held with its lower limbs biased by 2^31 (PADk) so that no limb goes
// negative; b limbs < 2^30, b < k*p.  Result normalised, value < a + k*p.
template <class WP> PLK_HD W9<WP> sub2(const W9<WP> &a, const W9<WP> &b) {
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + WP::PAD2[i] - b.l[i];
    return normw(r);
}
template <class WP> PLK_HD W9<WP> sub4(const W9<WP> &a, const W9<WP> &b) {
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + WP::PAD4[i] - b.l[i];
    return normw(r);
}
template <class WP> PLK_HD W9<WP> sub6(const W9<WP> &a, const W9<WP> &b) {
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + WP::PAD6[i] - b.l[i];
    return normw(r);
}
// 2p - a for a < 2p (negation)
template <class WP> PLK_HD W9<WP> neg2(const W9<WP> &a) {
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = WP::PAD2[i] - a.l[i];
    return normw(r);
}
template <class WP> PLK_HD W9<WP> addn(const W9<WP> &a, const W9<WP> &b) { return normw(addw(a, b)); }

// Montgomery product, radix 2^29, R' = 2^261 — PRODUCT SCANNING: the 18 columns of a*b + m*p are summed one after the
// other in ONE 64-bit accumulator; the carry of column k (acc >> 29) is the addend of the first multiply-add of column k+1,
// so no 64-bit addition is ever issued (operand scanning paid one v_lshl_add_u64 per row plus eight in the final
// normalisation: 58 non-mad instructions per product against 42 here).  CHAIN() pins that association: left alone, the
// compiler sums a column apart from the carry and joins the two with the very addition this form exists to avoid.
// Limbs: b < 2^29 (normalised); a may be an un-normalised sum or padded difference: a column receives at most 9 products
// a_j*b_i, 9 products m_i*p_j (both factors < 2^29) and one carry (< 2^35), so 9*A*(2^29-1) + 9*(2^29-1)^2 + 2^35 < 2^64
// allows a-limbs up to A = 3.28e9 (MULW_A_LIMB_MAX).  The sums the kernels feed in: x + y (< 2^30), x + PAD2 - y
// (< 2^29 + 2.66e9 = 3.19e9).  Checked on the host at the bound (tests/host/field29_check.hip).
#if defined(__HIP_DEVICE_COMPILE__)
#define PLK_CHAIN(acc) asm("" : "+v"(acc))
#else
#define PLK_CHAIN(acc) do { } while (0)
#endif
// the reduction half of a column: m_k for k < 9 (and its product with p_0), then the carry; or the output limb for k >= 9
#define PLK_MONT_LOW(k)  { _Pragma("unroll") for (int i = 0; i < (k); i++) { acc += (uint64_t)m[i] * WP::P29[(k) - i]; PLK_CHAIN(acc); } \
                           m[k] = ((uint32_t)acc * WP::INV29) & M29; acc += (uint64_t)m[k] * WP::P29[0]; acc >>= 29; }
#define PLK_MONT_HIGH(k) { _Pragma("unroll") for (int i = (k) - 8; i <= 8; i++) { acc += (uint64_t)m[i] * WP::P29[(k) - i]; PLK_CHAIN(acc); } \
                           r.l[(k) - 9] = (uint32_t)acc & M29; acc >>= 29; }
template <class WP>
PLK_HD W9<WP> mulw(const W9<WP> &a, const W9<WP> &b) {
    uint32_t m[9];
    W9<WP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)a.l[k - i] * b.l[i]; PLK_CHAIN(acc); }
        PLK_MONT_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)a.l[k - i] * b.l[i]; PLK_CHAIN(acc); }
        PLK_MONT_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
// Two independent products in lockstep: the multiply-adds of the two accumulator chains alternate, so that no
// v_mad_u64_u32 reads the result of the one issued just before it (on gfx950 that costs a wait state: the compiler
// puts an s_nop between every pair of a single chain).
#define PLK_MONT_LOW2(k)  { _Pragma("unroll") for (int i = 0; i < (k); i++) { acc0 += (uint64_t)m0[i] * WP::P29[(k) - i]; PLK_CHAIN(acc0); acc1 += (uint64_t)m1[i] * WP::P29[(k) - i]; PLK_CHAIN(acc1); } \
                            m0[k] = ((uint32_t)acc0 * WP::INV29) & M29; m1[k] = ((uint32_t)acc1 * WP::INV29) & M29; \
                            acc0 += (uint64_t)m0[k] * WP::P29[0]; acc1 += (uint64_t)m1[k] * WP::P29[0]; acc0 >>= 29; acc1 >>= 29; }
#define PLK_MONT_HIGH2(k) { _Pragma("unroll") for (int i = (k) - 8; i <= 8; i++) { acc0 += (uint64_t)m0[i] * WP::P29[(k) - i]; PLK_CHAIN(acc0); acc1 += (uint64_t)m1[i] * WP::P29[(k) - i]; PLK_CHAIN(acc1); } \
                            r0.l[(k) - 9] = (uint32_t)acc0 & M29; r1.l[(k) - 9] = (uint32_t)acc1 & M29; acc0 >>= 29; acc1 >>= 29; }
template <class WP>
PLK_HD void mulw2(const W9<WP> &a0, const W9<WP> &b0, const W9<WP> &a1, const W9<WP> &b1, W9<WP> &r0, W9<WP> &r1) {
    uint32_t m0[9], m1[9];
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc0 += (uint64_t)a0.l[k - i] * b0.l[i]; PLK_CHAIN(acc0); acc1 += (uint64_t)a1.l[k - i] * b1.l[i]; PLK_CHAIN(acc1); }
        PLK_MONT_LOW2(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc0 += (uint64_t)a0.l[k - i] * b0.l[i]; PLK_CHAIN(acc0); acc1 += (uint64_t)a1.l[k - i] * b1.l[i]; PLK_CHAIN(acc1); }
        PLK_MONT_HIGH2(k)
    }
    r0.l[8] = (uint32_t)acc0; r1.l[8] = (uint32_t)acc1;
}
// two independent squarings in lockstep (see mulw2)
template <class WP>
PLK_HD void sqrw2(const W9<WP> &x0, const W9<WP> &x1, W9<WP> &r0, W9<WP> &r1) {
    uint32_t m0[9], m1[9], d0[9], d1[9];
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) { d0[j] = x0.l[j] << 1; d1[j] = x1.l[j] << 1; }
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) { acc0 += (uint64_t)d0[k - i] * x0.l[i]; PLK_CHAIN(acc0); acc1 += (uint64_t)d1[k - i] * x1.l[i]; PLK_CHAIN(acc1); }
        if (k % 2 == 0) { acc0 += (uint64_t)x0.l[k / 2] * x0.l[k / 2]; PLK_CHAIN(acc0); acc1 += (uint64_t)x1.l[k / 2] * x1.l[k / 2]; PLK_CHAIN(acc1); }
        PLK_MONT_LOW2(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) { acc0 += (uint64_t)d0[k - i] * x0.l[i]; PLK_CHAIN(acc0); acc1 += (uint64_t)d1[k - i] * x1.l[i]; PLK_CHAIN(acc1); }
        if (k % 2 == 0) { acc0 += (uint64_t)x0.l[k / 2] * x0.l[k / 2]; PLK_CHAIN(acc0); acc1 += (uint64_t)x1.l[k / 2] * x1.l[k / 2]; PLK_CHAIN(acc1); }
        PLK_MONT_HIGH2(k)
    }
    r0.l[8] = (uint32_t)acc0; r1.l[8] = (uint32_t)acc1;
}
// a^2: the cross products a_i*a_j (i < j) are taken once against the doubled operand: 45 + 81 mads instead of 162
template <class WP>
PLK_HD W9<WP> sqrw(const W9<WP> &a) {
    uint32_t m[9], a2[9];
    W9<WP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) a2[j] = a.l[j] << 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) { acc += (uint64_t)a2[k - i] * a.l[i]; PLK_CHAIN(acc); }
        if (k % 2 == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; PLK_CHAIN(acc); }
        PLK_MONT_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) { acc += (uint64_t)a2[k - i] * a.l[i]; PLK_CHAIN(acc); }
        if (k % 2 == 0) { acc += (uint64_t)a.l[k / 2] * a.l[k / 2]; PLK_CHAIN(acc); }
        PLK_MONT_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// a*b + c*d with ONE Montgomery reduction (243 mads instead of 324).  Used as a*b - c*e by passing
// d = k*p - e.  Limbs: a, c < 2^30; b, d < 2^29.  Columns hold at most 9 * (2*2^59 + 2^58) < 2^63.4.
template <class WP>
PLK_HD W9<WP> mul2addw(const W9<WP> &a, const W9<WP> &b, const W9<WP> &c, const W9<WP> &d) {
    uint32_t m[9];
    W9<WP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (uint64_t)a.l[k - i] * b.l[i]; PLK_CHAIN(acc); acc += (uint64_t)c.l[k - i] * d.l[i]; PLK_CHAIN(acc); }
        PLK_MONT_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) { acc += (uint64_t)a.l[k - i] * b.l[i]; PLK_CHAIN(acc); acc += (uint64_t)c.l[k - i] * d.l[i]; PLK_CHAIN(acc); }
        PLK_MONT_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// ---- product by a CONSTANT that is held as three shifted copies ("tw3"): 108 multiply-adds instead of 162.
// For a constant w keep  W_q = w * 2^(87 (q+1)) mod p  (q = 0, 1, 2; canonical, 9 limbs each).  Split the variable operand into
// three chunks of three limbs, x = X_0 + X_1 2^87 + X_2 2^174: then  X_0 W_0 + X_1 W_1 + X_2 W_2 = x * w * 2^87 (mod p)  and all
// three partial products start at column 0, so only THREE Montgomery steps (one per limb of 2^87) are needed to divide the 2^87
// out again:
//     mul_tw3(x, W) = (X_0 W_0 + X_1 W_1 + X_2 W_2 + m p) / 2^87  =  x * w  (mod p),      81 + 27 multiply-adds, 12 columns.
// The constant costs 27 words instead of 9 — this is for constants that are read from a table many times (the NTT's stage
// twiddles); no domain change: the result is in whatever domain x is in, w is the PLAIN value of the constant.
// Bounds: a column holds at most 9 products x_j * W (x_j <= A, W < 2^29), 3 products m * p and a carry:
// 9 A (2^29 - 1) + 3 (2^29 - 1)^2 + 2^35 < 2^64 for A <= 3.6e9 (MULTW3_X_LIMB_MAX).  The result depends on the chunk values, not on
// the value of x: r < (X_0 + X_1 + X_2 + 2^87) p / 2^87, i.e. r < 4p for NORMALISED x (limbs < 2^29, so X_q < 2^87) — un-normalised
// limbs are allowed but buy a bound of up to 19p (limbs of 3.2e9), so callers normalise first.  Output limbs are normalised.
constexpr uint32_t MULTW3_X_LIMB_MAX = 3600000000u;
template <class WP> struct Tw3 { uint32_t w[3][9]; };
template <class WP>
PLK_HD W9<WP> mul_tw3(const W9<WP> &x, const Tw3<WP> &t) {
    uint32_t m[3];
    W9<WP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (k - i < 0 || k - i > 8) continue;
#pragma unroll
            for (int q = 0; q < 3; q++) { acc += (uint64_t)x.l[3 * q + i] * t.w[q][k - i]; PLK_CHAIN(acc); }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i >= k || k - i > 8) continue;
            acc += (uint64_t)m[i] * WP::P29[k - i]; PLK_CHAIN(acc);
        }
        if (k < 3) { m[k] = ((uint32_t)acc * WP::INV29) & M29; acc += (uint64_t)m[k] * WP::P29[0]; }
        else r.l[k - 3] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}
// two of them in lockstep with the SAME constant (the two products of a butterfly stage that share a twiddle)
template <class WP>
PLK_HD void mul_tw3_2(const W9<WP> &x0, const W9<WP> &x1, const Tw3<WP> &t, W9<WP> &r0, W9<WP> &r1) {
    uint32_t m0[3], m1[3];
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (k - i < 0 || k - i > 8) continue;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                acc0 += (uint64_t)x0.l[3 * q + i] * t.w[q][k - i]; PLK_CHAIN(acc0);
                acc1 += (uint64_t)x1.l[3 * q + i] * t.w[q][k - i]; PLK_CHAIN(acc1);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (i >= k || k - i > 8) continue;
            acc0 += (uint64_t)m0[i] * WP::P29[k - i]; PLK_CHAIN(acc0);
            acc1 += (uint64_t)m1[i] * WP::P29[k - i]; PLK_CHAIN(acc1);
        }
        if (k < 3) {
            m0[k] = ((uint32_t)acc0 * WP::INV29) & M29; m1[k] = ((uint32_t)acc1 * WP::INV29) & M29;
            acc0 += (uint64_t)m0[k] * WP::P29[0]; acc1 += (uint64_t)m1[k] * WP::P29[0];
        } else { r0.l[k - 3] = (uint32_t)acc0 & M29; r1.l[k - 3] = (uint32_t)acc1 & M29; }
        acc0 >>= 29; acc1 >>= 29;
    }
    r0.l[8] = (uint32_t)acc0; r1.l[8] = (uint32_t)acc1;
}
// the three copies of a constant given in the W domain (w * 2^261, canonical): W_2 is the value itself, W_1 and W_0 are it times
// 2^-87 and 2^-174 — products by the limb-unit vectors 2^174 and 2^87 (mulw divides by 2^261)
template <class WP>
PLK_HD Tw3<WP> make_tw3(const W9<WP> &w_dom_w) {
    W9<WP> e87 = w_zero<WP>(), e174 = w_zero<WP>();
    e87.l[3] = 1; e174.l[6] = 1;
    const W9<WP> w0 = csub_p(mulw(w_dom_w, e87)), w1 = csub_p(mulw(w_dom_w, e174));
    Tw3<WP> t;
    for (int i = 0; i < 9; i++) { t.w[0][i] = w0.l[i]; t.w[1][i] = w1.l[i]; t.w[2][i] = w_dom_w.l[i]; }
    return t;
}

// ---- OPERAND-SCANNING forms of the same three products (identical results, bit for bit): ten independent 64-bit
// column accumulators per row instead of one chain.  They issue 16 more instructions per product (the carries join their
// columns through v_lshl_add_u64), but a wave that has its SIMD to itself — the bucket-reduction kernels of the MSM are
// chains of dependent full additions run by one or two waves per SIMD — is bound by the dependent-issue latency of a single
// chain (on gfx950 a v_mad_u64_u32 feeding the next one costs a wait state), not by the instruction count: measured, the
// reduction kernels are 15-20 % faster on these forms (msm_task_reduce 0.25 -> 0.21 ms, msm_window_sums 0.13 -> 0.11 at
// 2^20 terms), while every throughput-bound kernel (NTT passes, point-wise kernels, the accumulation) is as fast or faster
// on the product-scanning forms above.
template <class WP>
PLK_HD W9<WP> mulw_os(const W9<WP> &a, const W9<WP> &b) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * b.l[i];
        const uint32_t m = ((uint32_t)t[0] * WP::INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * WP::P29[j];
        const uint64_t c = t[0] >> 29;                               // t[0] is now a multiple of 2^29
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[0] += c;
        t[9] = 0;
    }
    W9<WP> r;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t s = t[j] + c; r.l[j] = (uint32_t)s & M29; c = s >> 29; }
    r.l[8] = (uint32_t)(t[8] + c);
    return r;
}
// a^2: the cross products a_i*a_j (i < j) are taken once against the doubled operand, 45 + 81 mads instead
// of 162.  Row i adds a_i^2 to column 2i and 2*a_j*a_i (j > i) to column i+j; every product that belongs to
// global column g is in place before step g reduces it (the smaller index is <= g/2).
template <class WP>
PLK_HD W9<WP> sqrw_os(const W9<WP> &a) {
    uint64_t t[10];
    uint32_t a2[9];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) a2[j] = a.l[j] << 1;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        t[i] += (uint64_t)a.l[i] * a.l[i];                           // global column 2i = local column i
#pragma unroll
        for (int j = i + 1; j < 9; j++) t[j] += (uint64_t)a2[j] * a.l[i];   // global column i+j = local column j
        const uint32_t m = ((uint32_t)t[0] * WP::INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * WP::P29[j];
        const uint64_t c = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[0] += c;
        t[9] = 0;
    }
    W9<WP> r;
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t s = t[j] + c; r.l[j] = (uint32_t)s & M29; c = s >> 29; }
    r.l[8] = (uint32_t)(t[8] + c);
    return r;
}

// a*b + c*d with ONE Montgomery reduction (243 mads instead of 324).  Used as a*b - c*e by passing
// d = k*p - e.  Limbs: a, c < 2^30; b, d < 2^29.  Columns hold at most 9 * (2*2^59 + 2^58) < 2^63.4.
template <class WP>
PLK_HD W9<WP> mul2addw_os(const W9<WP> &a, const W9<WP> &b, const W9<WP> &c, const W9<WP> &d) {
    uint64_t t[10];
#pragma unroll
    for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)a.l[j] * b.l[i];
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)c.l[j] * d.l[i];
        const uint32_t m = ((uint32_t)t[0] * WP::INV29) & M29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] += (uint64_t)m * WP::P29[j];
        const uint64_t cy = t[0] >> 29;
#pragma unroll
        for (int j = 0; j < 9; j++) t[j] = t[j + 1];
        t[0] += cy;
        t[9] = 0;
    }
    W9<WP> r;
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t s = t[j] + cy; r.l[j] = (uint32_t)s & M29; cy = s >> 29; }
    r.l[8] = (uint32_t)(t[8] + cy);
    return r;
}

// exact conditional subtraction: normalised a < 2p  ->  a mod p in [0, p)
template <class WP>
PLK_HD W9<WP> csub_p(const W9<WP> &a) {
    int32_t d[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { int32_t s = (int32_t)a.l[i] - (int32_t)WP::P29[i] + c; d[i] = s & (int32_t)M29; c = s >> 29; }
    d[8] = (int32_t)a.l[8] - (int32_t)WP::P29[8] + c;
    const bool neg = d[8] < 0;
    W9<WP> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = neg ? a.l[i] : (uint32_t)d[i];
    return r;
}

// full reduction of any normalised value < 2^261 to the canonical residue: one product by 2^261 (the
// Montgomery "one" of this layer) brings it below ~1.01 p, then one exact subtraction
template <class WP>
PLK_HD W9<WP> reduce_full(const W9<WP> &a) { return csub_p(mulw(a, w_one<WP>())); }

// canonical residue of a normalised a < 64p without a product: q = floor(top limb / (top limb of p + 1)) is
// floor(a/p) or one less (the error term is below 1e-5), so a - q*p < 2p and one conditional subtraction ends it.
// ~100 cheap instructions against ~300 for the product by one.
template <class WP>
PLK_HD W9<WP> reduce_small(const W9<WP> &a) {
    const uint32_t q = a.l[8] / (WP::P29[8] + 1u);               // constant divisor: a multiply-high and a shift
    W9<WP> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int64_t s = (int64_t)a.l[i] - (int64_t)q * (int64_t)WP::P29[i] + c;
        r.l[i] = (uint32_t)s & M29;
        c = s >> 29;
    }
    r.l[8] = (uint32_t)((int64_t)a.l[8] - (int64_t)q * (int64_t)WP::P29[8] + c);
    return csub_p(r);
}

// a == 0 (mod p) for normalised a < 16p.  a = k*p forces l[0] * p^-1 = k (mod 2^29) with k < 16, which a
// random value passes with probability 2^-25; only then is the value reduced and compared.
template <class WP>
PLK_HD bool is_zero_mod_p(const W9<WP> &a) {
    uint32_t k = (a.l[0] * WP::PINV0) & M29;
    if (k >= 16) return false;
    return w_all_zero(reduce_full(a));
}

// the cheap half of the test above (no false negatives)
template <class WP>
PLK_HD bool maybe_zero_mod_p(const W9<WP> &a) { return ((a.l[0] * WP::PINV0) & M29) < 16; }

// a0*b0 + a1*b1 + a2*b2 with one reduction (3*81 + 81 mads instead of 3*162).  Normalised inputs; a column never
// holds more than 9 * (3 + 1) * 2^58 < 2^64.
template <class WP>
PLK_HD W9<WP> mulsum3w(const W9<WP> &a0, const W9<WP> &b0, const W9<WP> &a1, const W9<WP> &b1, const W9<WP> &a2, const W9<WP> &b2) {
    uint32_t m[9];
    W9<WP> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a0.l[k - i] * b0.l[i]; PLK_CHAIN(acc); acc += (uint64_t)a1.l[k - i] * b1.l[i]; PLK_CHAIN(acc);
            acc += (uint64_t)a2.l[k - i] * b2.l[i]; PLK_CHAIN(acc);
        }
        PLK_MONT_LOW(k)
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i <= 8; i++) {
            acc += (uint64_t)a0.l[k - i] * b0.l[i]; PLK_CHAIN(acc); acc += (uint64_t)a1.l[k - i] * b1.l[i]; PLK_CHAIN(acc);
            acc += (uint64_t)a2.l[k - i] * b2.l[i]; PLK_CHAIN(acc);
        }
        PLK_MONT_HIGH(k)
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// domain changes at the boundary of the W layer (s = packed external form, R = 2^256)
template <class WP> PLK_HD W9<WP> w_from_s(const W9<WP> &raw) { return mulw(raw, w_from_s_const<WP>()); }      // x*2^256 -> x*2^261 (< 1.1p)
template <class WP> PLK_HD W9<WP> s_from_w(const W9<WP> &a) { return csub_p(mulw(a, s_from_w_const<WP>())); }   // x*2^261 -> x*2^256 canonical

using FrW9 = W9<FrW>;
using FqW9 = W9<FqW>;

}  // namespace plk
