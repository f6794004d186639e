// G1 arithmetic over the 9 x 29-bit lazy field layer (field29_dev.h) — the MSM's inner loops.
// Same formulas as ec_dev.h (XYZZ coordinates: madd-2008-s, add-2008-s, dbl-2008-s-1), but every
// subtraction is a limb-wise "a - b + k*p" and nothing is conditionally reduced.  Invariants
// (values as multiples of p; all limbs normalised):
//     affine input       x, y      < 1.1 p     (resident SRS: canonical, 2^261 domain)
//     XYZZ accumulator   x, y      < 6 p       zz, zzz < 1.3 p
// Bound bookkeeping (out of mulw < a*b*0.0059/p + p):
//   madd:  U2,S2 < 1.02  P = U2-X+6p < 7.02  R = 8p-Y(+/-)S2 < 9.1  PP < 1.3  PPP < 1.06  Q < 1.05
//          X3 = R^2-PPP-2Q+4p < 5.5   T = Q-X3+6p < 7.05   Y3 = R*T + Y*(2p-PPP) (one reduction) < 1.5
//   add :  U,S < 1.05  P,R < 3.05  PP < 1.06  X3 < 5.06  Y3 < 3.2
//   dbl :  U = 2Y < 12  V < 1.85  W < 1.14  S < 1.07  M = 3X^2 < 3.7  X3 < 5.1  Y3 < 3.2
#pragma once
#include "field29_dev.h"
#include "ec_dev.h"

namespace plk {

struct AffW { FqW9 x, y; };
struct alignas(16) XyzzW { FqW9 x, y, zz, zzz; };              // 144 bytes

// the product is inlined: 162 v_mad_u64_u32 + ~60 other instructions (1.9 KB), ten of them per
// mixed addition keep the accumulate loop at ~25 KB, inside the 64 KB instruction cache
#define WM(a, b) mulw<FqW>((a), (b))
#define WS(a) sqrw<FqW>((a))                               // normalised input; 126 mads instead of 162
#define WMA(a, b, c, d) mul2addw<FqW>((a), (b), (c), (d))  // a*b + c*d, one reduction
// the latency-bound chains (full addition, doubling: bucket reduction, table build) take the operand-scanning forms
#define LM(a, b) mulw_os<FqW>((a), (b))
#define LS(a) sqrw_os<FqW>((a))
#define LMA(a, b, c, d) mul2addw_os<FqW>((a), (b), (c), (d))

PLK_HD bool is_inf(const XyzzW &p) { return w_all_zero(p.zz); }
PLK_HD bool is_inf(const AffW &p) { return w_all_zero(p.x) && w_all_zero(p.y); }
PLK_HD XyzzW xyzzw_identity() { XyzzW r; r.x = w_zero<FqW>(); r.y = w_zero<FqW>(); r.zz = w_zero<FqW>(); r.zzz = w_zero<FqW>(); return r; }

// 2 * (x, y) for an affine point; ysgn selects +y / -y
PLK_HD XyzzW xyzzw_double_affine(const FqW9 &x, const FqW9 &y) {
    FqW9 u = addn(y, y), v = WS(u), w = WM(u, v), s = WM(x, v);
    FqW9 xx = WS(x), m = normw(addw(addw(xx, xx), xx));
    XyzzW r;
    r.x = sub4(WS(m), addn(s, s));
    r.y = WMA(m, sub6(s, r.x), y, neg2(w));
    r.zz = v; r.zzz = w;
    return r;
}

PLK_HD XyzzW xyzzw_double(const XyzzW &p) {
    if (is_inf(p)) return p;
    FqW9 u = addn(p.y, p.y), v = LS(u), w = LM(u, v), s = LM(p.x, v);
    FqW9 xx = LS(p.x), m = normw(addw(addw(xx, xx), xx));
    XyzzW r;
    r.x = sub4(LS(m), addn(s, s));
    r.y = LMA(m, sub6(s, r.x), p.y, neg2(w));
    r.zz = LM(v, p.zz); r.zzz = LM(w, p.zzz);
    return r;
}

// slow path of the mixed addition: the x-difference passed the cheap zero filter.  Recomputes the
// exact tests; falls back to the generic addition when the filter was a false positive.
PLK_HD void xyzzw_add(XyzzW &a, const XyzzW &b);
// (inlined on purpose: a non-inlined callee is compiled with its own 200+ VGPR budget, which the
//  AMDGPU backend then charges to every kernel that can reach it — halving the occupancy of the hot loop)
PLK_HD void xyzzw_add_mixed_special(XyzzW &acc, const AffW &q, bool neg_q, const FqW9 &p, const FqW9 &r) {
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(r)) acc = xyzzw_double_affine(q.x, neg_q ? neg2(q.y) : q.y);
        else acc = xyzzw_identity();
        return;
    }
    XyzzW t; t.x = q.x; t.y = neg_q ? neg2(q.y) : q.y; t.zz = w_one<FqW>(); t.zzz = w_one<FqW>();
    xyzzw_add(acc, t);
}

// acc += (qx, +-qy)
PLK_HD void xyzzw_add_mixed(XyzzW &acc, const AffW &q, bool neg_q) {
    if (is_inf(q)) return;
    if (is_inf(acc)) { acc.x = q.x; acc.y = neg_q ? neg2(q.y) : q.y; acc.zz = w_one<FqW>(); acc.zzz = w_one<FqW>(); return; }
    FqW9 u2, s2;
    mulw2<FqW>(q.x, acc.zz, q.y, acc.zzz, u2, s2);
    FqW9 p = sub6(u2, acc.x);
    FqW9 r;
    {
        const uint32_t m = neg_q ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = FqW::PAD8[i] - acc.y.l[i] + ((s2.l[i] ^ m) - m);
        r = normw(r);
    }
    if (maybe_zero_mod_p(p)) {                               // P == +-Q: rare, kept out of the hot code
        xyzzw_add_mixed_special(acc, q, neg_q, p, r);
        return;
    }
    FqW9 pp, rr, ppp, qq;
    sqrw2<FqW>(p, r, pp, rr);
    mulw2<FqW>(p, pp, acc.x, pp, ppp, qq);
    FqW9 x3;
    {
#pragma unroll
        for (int i = 0; i < 9; i++) x3.l[i] = rr.l[i] + FqW::PAD4[i] - ppp.l[i] - 2 * qq.l[i];
        x3 = normw(x3);
    }
    FqW9 zz3, zzz3;
    mulw2<FqW>(acc.zz, pp, acc.zzz, ppp, zz3, zzz3);
    acc.y = WMA(r, sub6(qq, x3), acc.y, neg2(ppp));           // R*(Q - X3) - Y*PPP, one reduction
    acc.x = x3;
    acc.zz = zz3;
    acc.zzz = zzz3;
}

// a += b
PLK_HD void xyzzw_add(XyzzW &a, const XyzzW &b) {
    if (is_inf(b)) return;
    if (is_inf(a)) { a = b; return; }
    FqW9 u1 = LM(a.x, b.zz), u2 = LM(b.x, a.zz), s1 = LM(a.y, b.zzz), s2 = LM(b.y, a.zzz);
    FqW9 p = sub2(u2, u1), r = sub2(s2, s1);
    if (is_zero_mod_p(p)) {
        if (is_zero_mod_p(r)) a = xyzzw_double(a);
        else a = xyzzw_identity();
        return;
    }
    FqW9 pp = LS(p), ppp = LM(p, pp), qq = LM(u1, pp);
    FqW9 x3;
    {
        FqW9 rr = LS(r);
#pragma unroll
        for (int i = 0; i < 9; i++) x3.l[i] = rr.l[i] + FqW::PAD4[i] - ppp.l[i] - 2 * qq.l[i];
        x3 = normw(x3);
    }
    a.y = LMA(r, sub6(qq, x3), s1, neg2(ppp));
    a.x = x3;
    a.zz = LM(LM(a.zz, b.zz), pp);
    a.zzz = LM(LM(a.zzz, b.zzz), ppp);
}

// storage <-> registers
__device__ __forceinline__ AffW load_affw(const G1Affine *p) {            // packed, already in the 2^261 domain
    AffW r; r.x = unpack<FqW>(load_fp(&p->x)); r.y = unpack<FqW>(load_fp(&p->y)); return r;
}
__device__ __forceinline__ FqW9 load_w(const uint32_t *p) {
    FqW9 r;
    const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
    u32x4 a = q[0], b = q[1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w; r.l[8] = p[8];
    return r;
}
__device__ __forceinline__ void store_w(uint32_t *p, const FqW9 &v) {
    u32x4 *q = reinterpret_cast<u32x4 *>(p);
    q[0] = u32x4{v.l[0], v.l[1], v.l[2], v.l[3]};
    q[1] = u32x4{v.l[4], v.l[5], v.l[6], v.l[7]};
    p[8] = v.l[8];
}
// XyzzW in memory: 4 coordinates x (8 limbs in two 16-byte words + the 9th in a trailing block) = 144 B
__device__ __forceinline__ XyzzW load_xyzzw(const XyzzW *p) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
    XyzzW r;
    FqW9 *c[4] = {&r.x, &r.y, &r.zz, &r.zzz};
    u32x4 tail = *reinterpret_cast<const u32x4 *>(w + 32);
    const uint32_t t[4] = {tail.x, tail.y, tail.z, tail.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32x4 *q = reinterpret_cast<const u32x4 *>(w + 8 * k);
        u32x4 a = q[0], b = q[1];
        c[k]->l[0] = a.x; c[k]->l[1] = a.y; c[k]->l[2] = a.z; c[k]->l[3] = a.w;
        c[k]->l[4] = b.x; c[k]->l[5] = b.y; c[k]->l[6] = b.z; c[k]->l[7] = b.w; c[k]->l[8] = t[k];
    }
    return r;
}
__device__ __forceinline__ void store_xyzzw(XyzzW *p, const XyzzW &v) {
    uint32_t *w = reinterpret_cast<uint32_t *>(p);
    const FqW9 *c[4] = {&v.x, &v.y, &v.zz, &v.zzz};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32x4 *q = reinterpret_cast<u32x4 *>(w + 8 * k);
        q[0] = u32x4{c[k]->l[0], c[k]->l[1], c[k]->l[2], c[k]->l[3]};
        q[1] = u32x4{c[k]->l[4], c[k]->l[5], c[k]->l[6], c[k]->l[7]};
    }
    *reinterpret_cast<u32x4 *>(w + 32) = u32x4{v.x.l[8], v.y.l[8], v.zz.l[8], v.zzz.l[8]};
}

// XYZZ (2^261 domain, lazy) -> XYZZ in the library's external form (canonical, R = 2^256), for the host
__device__ __forceinline__ G1Xyzz xyzzw_export(const XyzzW &p) {
    G1Xyzz r;
    if (is_inf(p)) return xyzz_identity();
    r.x = pack<FqParams>(s_from_w(p.x)); r.y = pack<FqParams>(s_from_w(p.y));
    r.zz = pack<FqParams>(s_from_w(p.zz)); r.zzz = pack<FqParams>(s_from_w(p.zzz));
    return r;
}

}  // namespace plk
