// One XYZZ addition computed by FOUR lanes (a quad: lanes 4q .. 4q+3 of a wave) — for the latency-bound reduction trees.
//
// A full addition (add-2008-s, ec29_dev.h: xyzzw_add) is 14 field products; issued by one lane's wave they are a dependent chain of
// ~7 us on a SIMD the wave has to itself, and every level of a reduction tree pays that chain.  The 14 products are only FOUR deep:
//     stage 1   U1 = X1 ZZ2     U2 = X2 ZZ1      S1 = Y1 ZZZ2       S2 = Y2 ZZZ1        P = U2 - U1,  R = S2 - S1
//     stage 2   PP = P^2       RR = R^2         ZZ12 = ZZ1 ZZ2     ZZZ12 = ZZZ1 ZZZ2
//     stage 3   PPP = P PP     Q = U1 PP        ZZ3 = ZZ12 PP      —                    X3 = RR - PPP - 2 Q
//     stage 4   A = R (Q - X3) B = S1 PPP       —                  ZZZ3 = ZZZ12 PPP     Y3 = A - B
// so the four lanes of a quad take one product of a stage each.  First version (first half of round 6; in the history at d165fbe^): every lane held BOTH
// operands in full and selected its product's operands — ~550 selects and DPP moves per addition, 5.3 us per tree level.  This file is the second version,
// the DISTRIBUTED form below: 2.6 us per level for a wave that has its SIMD to itself.  Same value bounds as xyzzw_add (ec29_dev.h) except
// Y3 = A + 2p - B < 3.2 p (two reductions instead of the fused one).  Special cases: an operand at infinity is a select at the end; P = 0 (mod p) — doubling,
// or a point and its opposite — is rare and leaves through the ordinary lane-wise addition on gathered operands.
// Checked on the device against the lane-wise addition: tests/host/quad_add_check.hip (tests/test_gpu_quad_add.py).
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

// DISTRIBUTED form: lane r of a quad holds ONLY coordinate r of a point (0: X, 1: Y, 2: ZZ, 3: ZZZ) — 9 registers per operand
// instead of 36, and no four-way operand selects: the same four product stages as above with the values placed so that most operands
// are already where they are needed.  A and B in, A + B out, all in this form:
//     B' = B by quad_perm [2,3,0,1]                   lane:     0          1          2          3
//     T1 = A * B'                                               U1         S1         U2         S2
//     D  = T1 by quad_perm [2,3,0,1] - T1                       P          R          (-P)       (-R)
//     T2 = (lanes 0,1: D * D; lanes 2,3: A * B)                 PP         RR         ZZ12       ZZZ12
//     T3 = (D | - | T2 | U1 from lane 0) * PP from lane 0       PPP        -          ZZ3        Q
//     X3 = RR - PPP - 2 Q (every lane, from three broadcasts)
//     T4 = (S1 from lane 1 | D | - | T2) * (PPP | Q - X3 | - | PPP)   B    A          -          ZZZ3
//     out = X3 | A - B | T3 | T4
// ~80 DPP moves and ~110 selects per addition instead of ~550, and a partner's point is 9 ds_bpermute instead of 36 (1310 instructions in all, 648 of them
// multiply-adds).  The Montgomery product does not depend on the order of its operands, so which lane forms which product changes no value.
constexpr int QP_SWAP2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);            // quad_perm:[2,3,0,1]
template <int CTRL>
__device__ __forceinline__ FqW9 quad_dpp(const FqW9 &v) {
    FqW9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)v.l[i], CTRL, 0xf, 0xf, false);
    return r;
}
template <int SRC>
__device__ __forceinline__ bool quad_flag(bool f) { return __builtin_amdgcn_mov_dpp((int)f, SRC * 0x55, 0xf, 0xf, false) != 0; }
__device__ __forceinline__ FqW9 wsel(bool c, const FqW9 &a, const FqW9 &b) {
    FqW9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}
// coordinate `role` of a point in memory / of a point held in full by this lane
__device__ __forceinline__ FqW9 load_coord(const XyzzW *p, uint32_t role) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    const u32x4 a = *reinterpret_cast<const u32x4 *>(w + 8 * role), b = *reinterpret_cast<const u32x4 *>(w + 8 * role + 4);
    FqW9 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = w[32 + role];
    return r;
}
__device__ __forceinline__ void store_coord(XyzzW *p, uint32_t role, const FqW9 &v) {
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    *reinterpret_cast<u32x4 *>(w + 8 * role) = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
    *reinterpret_cast<u32x4 *>(w + 8 * role + 4) = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
    w[32 + role] = v.l[8];
}
__device__ __forceinline__ FqW9 coord_shfl_xor(const FqW9 &v, int mask) {      // (mask a multiple of 4: the partner quad's lane of the same role)
    FqW9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = __shfl_xor(v.l[i], mask);
    return r;
}
// the point held in full by lane SRC of the quad, in distributed form
template <int SRC>
__device__ __forceinline__ FqW9 quad_distribute(const XyzzW &v, uint32_t role) {
    const FqW9 x = quad_bcast<SRC>(v.x), y = quad_bcast<SRC>(v.y), zz = quad_bcast<SRC>(v.zz), zzz = quad_bcast<SRC>(v.zzz);
    return wsel(role < 2, wsel(role == 0, x, y), wsel(role == 2, zz, zzz));
}
// distributed -> held in full by every lane of the quad
__device__ __forceinline__ XyzzW quad_gather(const FqW9 &c) {
    XyzzW r; r.x = quad_bcast<0>(c); r.y = quad_bcast<1>(c); r.zz = quad_bcast<2>(c); r.zzz = quad_bcast<3>(c);
    return r;
}

__device__ __forceinline__ FqW9 xyzzw_add_dist(const FqW9 &A, const FqW9 &B, uint32_t role) {
    const bool inf_a = quad_flag<2>(w_all_zero(A)), inf_b = quad_flag<2>(w_all_zero(B));
    const FqW9 T1 = LM(A, quad_dpp<QP_SWAP2>(B));                                   // U1 | S1 | U2 | S2
    const FqW9 D = sub2(quad_dpp<QP_SWAP2>(T1), T1);                                // P | R | . | .
    if (!inf_a && !inf_b && quad_flag<0>(maybe_zero_mod_p(D))) {                     // (quad-uniform) doubling, opposite points or a false alarm: rare
        XyzzW a = quad_gather(A);
        const XyzzW b = quad_gather(B);
        xyzzw_add(a, b);
        return wsel(role < 2, wsel(role == 0, a.x, a.y), wsel(role == 2, a.zz, a.zzz));
    }
    const bool low = role < 2;
    const FqW9 T2 = LM(wsel(low, D, A), wsel(low, D, B));                           // PP | RR | ZZ12 | ZZZ12
    const FqW9 pp = quad_bcast<0>(T2);
    const FqW9 T3 = LM(wsel(role == 0, D, wsel(role == 2, T2, quad_bcast<0>(T1))), pp);   // PPP | . | ZZ3 | Q
    const FqW9 rr = quad_bcast<1>(T2), ppp = quad_bcast<0>(T3), qq = quad_bcast<3>(T3);
    FqW9 x3;
#pragma unroll
    for (int i = 0; i < 9; i++) x3.l[i] = rr.l[i] + FqW::PAD4[i] - ppp.l[i] - 2 * qq.l[i];
    x3 = normw(x3);
    const FqW9 rhs = sub6(qq, x3);
    const FqW9 T4 = LM(wsel(role == 1, D, wsel(role == 3, T2, quad_bcast<1>(T1))), wsel(role == 1, rhs, ppp));   // B | A | . | ZZZ3
    const FqW9 y3 = sub2(T4, quad_bcast<0>(T4));
    const FqW9 out = wsel(low, wsel(role == 0, x3, y3), wsel(role == 2, T3, T4));
    return wsel(inf_b, A, wsel(inf_a, B, out));
}

}  // namespace plk
