// One XYZZ addition computed by FOUR lanes (a quad: lanes 4q .. 4q+3 of a wave) — for the latency-bound reduction trees.
//
// A full addition (add-2008-s, ec29_dev.h: xyzzw_add) is 14 field products; issued by one lane's wave they are a dependent chain of
// ~7 us on a SIMD the wave has to itself, and every level of a reduction tree pays that chain — the trees behind a short commitment
// (msm_small.hip: 15 levels) were 110 us of its 165.  The 14 products are only FOUR deep:
//     stage 1   U1 = X1 ZZ2     U2 = X2 ZZ1      S1 = Y1 ZZZ2       S2 = Y2 ZZZ1        P = U2 - U1,  R = S2 - S1
//     stage 2   PP = P^2       RR = R^2         ZZ12 = ZZ1 ZZ2     ZZZ12 = ZZZ1 ZZZ2
//     stage 3   PPP = P PP     Q = U1 PP        ZZ3 = ZZ12 PP      —                    X3 = RR - PPP - 2 Q
//     stage 4   A = R (Q - X3) B = S1 PPP       —                  ZZZ3 = ZZZ12 PPP     Y3 = A - B
// so the four lanes of a quad hold BOTH operands in full, pick one product of the stage each by their lane number (v_cndmask), and
// hand the results round with DPP quad_perm moves (no LDS, no wait): four products + ~280 moves / selects per addition instead of
// fourteen products.  A wave performs 16 such additions at a time; the result is left in all four lanes.  Same value bounds as
// xyzzw_add (ec29_dev.h) except Y3 = A + 2p - B < 3.2 p (two reductions instead of the fused one).
// Special cases: an operand at infinity is a select at the end; P = 0 (mod p) — doubling, or a point and its opposite — is rare and
// leaves through the ordinary single-lane addition, computed by the four lanes redundantly.
#pragma once
#include "ec29_dev.h"

namespace plk {

template <int SRC>
__device__ __forceinline__ FqW9 quad_bcast(const FqW9 &v) {
    FqW9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], SRC * 0x55, 0xf, 0xf, false);   // quad_perm:[SRC,SRC,SRC,SRC]
    return r;
}
// the operand of lane `role`: two levels of bit-field selects by the two bits of the lane number (v_bfi_b32: three instructions per limb, no
// control flow — written as conditional expressions the selects became branches over pointers to the operands, which put them in scratch memory)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t on, uint32_t off) { return (on & mask) | (off & ~mask); }
__device__ __forceinline__ FqW9 quad_sel(uint32_t role, const FqW9 &v0, const FqW9 &v1, const FqW9 &v2, const FqW9 &v3) {
    const uint32_t m0 = 0u - (role & 1u), m1 = 0u - (role >> 1);
    FqW9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = bfi(m1, bfi(m0, v3.l[i], v2.l[i]), bfi(m0, v1.l[i], v0.l[i]));
    return r;
}

// a + b; every lane of the quad passes the same a and the same b and receives the same sum
__device__ __forceinline__ XyzzW xyzzw_add_quad(const XyzzW &a, const XyzzW &b, uint32_t role) {
    const bool inf_a = is_inf(a), inf_b = is_inf(b);
    const FqW9 t1 = LM(quad_sel(role, a.x, b.x, a.y, b.y), quad_sel(role, b.zz, a.zz, b.zzz, a.zzz));
    const FqW9 u1 = quad_bcast<0>(t1), u2 = quad_bcast<1>(t1), s1 = quad_bcast<2>(t1), s2 = quad_bcast<3>(t1);
    const FqW9 p = sub2(u2, u1), r = sub2(s2, s1);
    if (!inf_a && !inf_b && maybe_zero_mod_p(p)) {             // (quad-uniform: all four lanes hold the same values)
        XyzzW t = a;
        xyzzw_add(t, b);
        return t;
    }
    const FqW9 t2 = LM(quad_sel(role, p, r, a.zz, a.zzz), quad_sel(role, p, r, b.zz, b.zzz));    // PP | RR | ZZ12 | ZZZ12
    const FqW9 pp = quad_bcast<0>(t2), rr = quad_bcast<1>(t2);
    const FqW9 t3 = LM(quad_sel(role, p, u1, t2, t2), pp);                                         // PPP | Q | ZZ3 | (unused)
    const FqW9 ppp = quad_bcast<0>(t3), qq = quad_bcast<1>(t3);
    XyzzW o;
    {
        FqW9 x3;
#pragma unroll
        for (int i = 0; i < 9; i++) x3.l[i] = rr.l[i] + FqW::PAD4[i] - ppp.l[i] - 2 * qq.l[i];
        o.x = normw(x3);
    }
    FqW9 rhs = sub6(qq, o.x);
    {
        const uint32_t first = 0u - (uint32_t)(role == 0);
#pragma unroll
        for (int i = 0; i < 9; i++) rhs.l[i] = bfi(first, rhs.l[i], ppp.l[i]);
    }
    const FqW9 t4 = LM(quad_sel(role, r, s1, t2, t2), rhs);                                        // A | B | (unused) | ZZZ3
    o.y = sub2(quad_bcast<0>(t4), quad_bcast<1>(t4));
    o.zz = quad_bcast<2>(t3);
    o.zzz = quad_bcast<3>(t4);
    const uint32_t ka = 0u - (uint32_t)inf_b, kb = 0u - (uint32_t)(inf_a && !inf_b);       // keep a / keep b
    auto pick = [&](const FqW9 &g, const FqW9 &va, const FqW9 &vb) __attribute__((always_inline)) {
        FqW9 r;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = bfi(ka, va.l[i], bfi(kb, vb.l[i], g.l[i]));
        return r;
    };
    XyzzW out;
    out.x = pick(o.x, a.x, b.x); out.y = pick(o.y, a.y, b.y); out.zz = pick(o.zz, a.zz, b.zz); out.zzz = pick(o.zzz, a.zzz, b.zzz);
    return out;
}

}  // namespace plk
