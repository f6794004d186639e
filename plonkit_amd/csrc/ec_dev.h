// BN254 G1 (y^2 = x^3 + 3 over Fq) group arithmetic for the MSM / G1-iNTT kernels.
// Replaces pairing_ce::bn256::{G1Affine, G1} as used by bellman_ce::multiexp (reference call
// sites: commitments in prove / make_verification_key, src/plonk.rs:122-124,132-176).
//
// Device accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed addition is 8M + 2S against 7M + 4S for Jacobian, and with every multiply costing the
// same ~130 v_mad_u64_u32 on gfx950 the count of products is what matters.  Infinity: ZZ == 0.
// Affine points: (x, y) Montgomery, infinity encoded as x == y == 0 (as in the C ABI).
#pragma once
#include "field_dev.h"

namespace plk {

// The EC formulas below call the Montgomery product through ONE out-of-line copy on the device:
// fully inlined, a mixed addition is ~45 KB of straight-line code (10 products x ~560 instructions)
// and the accumulate loop thrashes the 64 KB instruction cache shared by two CUs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __noinline__ Fq fq_mul_call(Fq a, Fq b) { return plk::mul<FqParams>(a, b); }
#define ECM(a, b) fq_mul_call((a), (b))
#define ECS(a) fq_mul_call((a), (a))
#else
#define ECM(a, b) plk::mul((a), (b))
#define ECS(a) plk::mul((a), (a))
#endif

struct alignas(16) G1Affine { Fq x, y; };
struct alignas(16) G1Xyzz { Fq x, y, zz, zzz; };
struct alignas(16) G1Jac { Fq x, y, z; };

PLK_HD bool is_inf(const G1Affine &p) { return p.x.is_zero() && p.y.is_zero(); }
PLK_HD bool is_inf(const G1Xyzz &p) { return p.zz.is_zero(); }
PLK_HD G1Xyzz xyzz_identity() { G1Xyzz r; r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero(); return r; }
PLK_HD G1Xyzz xyzz_from_affine(const G1Affine &p) {
    if (is_inf(p)) return xyzz_identity();
    G1Xyzz r; r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one(); return r;
}

// 2 * affine  (mdbl-2008-s-1)
PLK_HD G1Xyzz xyzz_double_affine(const G1Affine &p) {
    Fq u = dbl(p.y), v = ECS(u), w = ECM(u, v), s = ECM(p.x, v);
    Fq xx = ECS(p.x), m = add(dbl(xx), xx);
    G1Xyzz r;
    r.x = sub(ECS(m), dbl(s));
    r.y = sub(ECM(m, sub(s, r.x)), ECM(w, p.y));
    r.zz = v; r.zzz = w;
    return r;
}

// 2 * xyzz  (dbl-2008-s-1)
PLK_HD G1Xyzz xyzz_double(const G1Xyzz &p) {
    if (is_inf(p)) return p;
    Fq u = dbl(p.y), v = ECS(u), w = ECM(u, v), s = ECM(p.x, v);
    Fq xx = ECS(p.x), m = add(dbl(xx), xx);
    G1Xyzz r;
    r.x = sub(ECS(m), dbl(s));
    r.y = sub(ECM(m, sub(s, r.x)), ECM(w, p.y));
    r.zz = ECM(v, p.zz); r.zzz = ECM(w, p.zzz);
    return r;
}

// acc += q (affine; negated when neg)   (madd-2008-s)
PLK_HD void xyzz_add_mixed(G1Xyzz &acc, const G1Affine &q_in, bool neg_q) {
    if (is_inf(q_in)) return;
    G1Affine q = q_in;
    if (neg_q) q.y = neg(q.y);
    if (is_inf(acc)) { acc.x = q.x; acc.y = q.y; acc.zz = Fq::one(); acc.zzz = Fq::one(); return; }
    Fq u2 = ECM(q.x, acc.zz), s2 = ECM(q.y, acc.zzz);
    Fq p = sub(u2, acc.x), r = sub(s2, acc.y);
    if (p.is_zero()) {
        if (r.is_zero()) acc = xyzz_double_affine(q);
        else acc = xyzz_identity();
        return;
    }
    Fq pp = ECS(p), ppp = ECM(p, pp), qq = ECM(acc.x, pp);
    Fq x3 = sub(sub(ECS(r), ppp), dbl(qq));
    acc.y = sub(ECM(r, sub(qq, x3)), ECM(acc.y, ppp));
    acc.x = x3;
    acc.zz = ECM(acc.zz, pp);
    acc.zzz = ECM(acc.zzz, ppp);
}

// a += b   (add-2008-s)
PLK_HD void xyzz_add(G1Xyzz &a, const G1Xyzz &b) {
    if (is_inf(b)) return;
    if (is_inf(a)) { a = b; return; }
    Fq u1 = ECM(a.x, b.zz), u2 = ECM(b.x, a.zz), s1 = ECM(a.y, b.zzz), s2 = ECM(b.y, a.zzz);
    Fq p = sub(u2, u1), r = sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) a = xyzz_double(a);
        else a = xyzz_identity();
        return;
    }
    Fq pp = ECS(p), ppp = ECM(p, pp), qq = ECM(u1, pp);
    Fq x3 = sub(sub(ECS(r), ppp), dbl(qq));
    a.y = sub(ECM(r, sub(qq, x3)), ECM(s1, ppp));
    a.x = x3;
    a.zz = ECM(ECM(a.zz, b.zz), pp);
    a.zzz = ECM(ECM(a.zzz, b.zzz), ppp);
}

PLK_HD G1Xyzz xyzz_neg(const G1Xyzz &p) { G1Xyzz r = p; r.y = neg(p.y); return r; }

// k * p for a small non-negative integer k (double-and-add, MSB first)
PLK_HD G1Xyzz xyzz_mul_small(const G1Xyzz &p, uint32_t k) {
    G1Xyzz acc = xyzz_identity();
    for (int i = 31; i >= 0; i--) {
        acc = xyzz_double(acc);
        if ((k >> i) & 1) xyzz_add(acc, p);
    }
    return acc;
}

// XYZZ -> Jacobian with Z = ZZ*ZZZ:  X' = X*ZZ*ZZZ^2, Y' = Y*ZZ^3*ZZZ^2
PLK_HD G1Jac xyzz_to_jacobian(const G1Xyzz &p) {
    G1Jac r;
    if (is_inf(p)) { r.x = Fq::one(); r.y = Fq::one(); r.z = Fq::zero(); return r; }
    Fq t2 = ECS(p.zzz), zz2 = ECS(p.zz);
    r.x = ECM(ECM(p.x, p.zz), t2);
    r.y = ECM(ECM(ECM(p.y, zz2), p.zz), t2);
    r.z = ECM(p.zz, p.zzz);
    return r;
}

__device__ __forceinline__ G1Affine load_affine(const G1Affine *p) {
    G1Affine r; r.x = load_fp(&p->x); r.y = load_fp(&p->y); return r;
}
__device__ __forceinline__ G1Xyzz load_xyzz(const G1Xyzz *p) {
    G1Xyzz r; r.x = load_fp(&p->x); r.y = load_fp(&p->y); r.zz = load_fp(&p->zz); r.zzz = load_fp(&p->zzz); return r;
}
__device__ __forceinline__ void store_xyzz(G1Xyzz *p, const G1Xyzz &v) {
    store_fp(&p->x, v.x); store_fp(&p->y, v.y); store_fp(&p->zz, v.zz); store_fp(&p->zzz, v.zzz);
}

}  // namespace plk
