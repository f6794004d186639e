// Inverse NTT over G1 — `plonkit dump-lagrange`:
//   Crs::<Bn256, CrsForLagrangeForm>::from_powers(&mono, n.next_power_of_two(), &Worker)
//   (src/plonk.rs:179-185; driven from src/bin/main.rs:360-381).
// in[j] = tau^j * G  ->  out[i] = L_i(tau) * G, the Lagrange-basis SRS of the size-N domain.
// Radix-2 DIT over group elements: a butterfly is (A, B) -> (A + w*B, A - w*B) where w*B is a full 254-bit scalar
// multiplication, so the kernel is bound by v_mad_u64_u32 issue and HBM traffic (144 B per point per stage) is
// negligible.  One lane per butterfly, XYZZ coordinates on the 9 x 29-bit layer between stages, a single Fermat
// inversion per point at the end; 1/N rides on the last stage (N/2 extra scalar multiplications, not N).
//
// Scalar multiplication on a SIMT machine: with double-and-add (or any sparse recoding) some lane of the wave has a
// non-zero digit at nearly every bit, so the whole wave pays one addition per bit.  Fixed signed 3-bit windows make
// all lanes add at the same positions; the window table {1,2,3,4}*B of every lane lives in LDS, limb-major
// (conflict-free).  Round 1: 255 doublings + 85 additions (~3500 field products).  Round 4: the GLV endomorphism halves
// the doubling chain — 129 doublings + 86 additions + 43 products by beta (~2400), see g1_mul_scalar / glv_dev.h.  Round 6: the window table made
// effectively affine on an isomorphic curve (eight points, 4-bit windows, mixed additions): ~1960, see g1_mul_scalar_iso8.
#include "ctx.h"
#include "ec_dev.h"
#include "ec29_dev.h"
#include "glv_dev.h"
#include "ntt.h"
#include <type_traits>

namespace plk {

constexpr int G1NTT_THREADS = 256;
constexpr int G1NTT_TABLE = 4;                                     // |digit| <= 4
constexpr size_t G1NTT_LDS = (size_t)G1NTT_TABLE * 36 * G1NTT_THREADS * sizeof(uint32_t);   // 147456 B: one workgroup per CU

__device__ __forceinline__ uint32_t brev32(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// rarely executed additions / doublings go through one out-of-line copy each (code size); the loop has its own inlined sites
__device__ __noinline__ void g1_add_call(XyzzW *a, const XyzzW *b) { XyzzW t = *a; xyzzw_add(t, *b); *a = t; }
__device__ __noinline__ void g1_double_call(XyzzW *a) { XyzzW t = *a; *a = xyzzw_double(t); }

__device__ __forceinline__ void lds_put(uint32_t *tab, int e, const XyzzW &p) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 36; k++) tab[(e * 36 + k) * G1NTT_THREADS + threadIdx.x] = w[k];
}
__device__ __forceinline__ XyzzW lds_get(const uint32_t *tab, int e) {
    XyzzW p;
    uint32_t *w = reinterpret_cast<uint32_t *>(&p);
#pragma unroll
    for (int k = 0; k < 36; k++) w[k] = tab[(e * 36 + k) * G1NTT_THREADS + threadIdx.x];
    return p;
}

// k * b for a canonical (non-Montgomery) scalar k < r, by the GLV endomorphism (glv_dev.h): k = k1 + k2 lambda with
// |k1|, |k2| < 2^128 and lambda * (x, y) = (beta x, y), so both halves share ONE chain of 129 doublings:
//     acc <- 8 acc ; acc += d1_w * b ; acc += d2_w * phi(b)          for the 43 signed 3-bit windows, high to low,
// d in [-3, 4] as in round 1, phi of a table entry = its x times beta (one more product).  129 doublings + 86 additions + 43
// products by beta = ~2400 field products per multiplication instead of the ~3500 of 255 doublings + 85 additions.  The two
// additions of a window go through ONE inlined addition site (a two-trip loop that is not unrolled: instruction cache).
__device__ __forceinline__ XyzzW g1_mul_scalar(const XyzzW &b, const Fr &k, uint32_t *tab) {
    if (is_inf(b)) return xyzzw_identity();
    {   // table: b, 2b, 3b, 4b
        XyzzW t = b, t2 = b;
        lds_put(tab, 0, t);
        g1_double_call(&t2);
        lds_put(tab, 1, t2);
        t = t2; g1_add_call(&t, &b);
        lds_put(tab, 2, t);
        g1_double_call(&t2);
        lds_put(tab, 3, t2);
    }
    uint32_t dig[2][6];
    uint32_t flip[2];                                             // sign of the half: XORed into the digit's sign bit
    {
        const GlvSplit sp = glv_split(k.l);
        glv_digits(sp.k1, dig[0]);
        glv_digits(sp.k2, dig[1]);
        flip[0] = sp.neg1 ? 8u : 0u; flip[1] = sp.neg2 ? 8u : 0u;
    }
    FqW9 beta;
#pragma unroll
    for (int i = 0; i < 9; i++) beta.l[i] = glv::BETA_W[i];
    XyzzW acc = xyzzw_identity();
    for (int w = 42; w >= 0; w--) {
        for (int r = 0; r < 3; r++) acc = xyzzw_double(acc);      // one inlined doubling site (identity passes through)
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t code = ((dig[h][w >> 3] >> (4 * (w & 7))) & 15u), mag = code & 7u;
            XyzzW t = xyzzw_identity();
            if (mag) {
                t = lds_get(tab, (int)mag - 1);
                if (h) t.x = WM(t.x, beta);                        // phi: x -> beta x (entries are normalised, x < 6p -> < 1.04p)
                if ((code ^ flip[h]) & 8u) t.y = sub6(w_zero<FqW>(), t.y);     // 6p - y: y < 6p by the bounds of ec29_dev.h
            }
            xyzzw_add(acc, t);                                    // the one inlined addition site (identity operand: no-op)
        }
    }
    return acc;
}

// ---- the same multiplication with an EFFECTIVELY AFFINE window table (late round 6).  The four table points b, 2b, 3b, 4b are brought to ONE
// denominator pair (D_zz = prod ZZ_i, D_zzz = prod ZZZ_i: x_i = X_i' / D_zz, y_i = Y_i' / D_zzz with X_i' = X_i prod_{j != i} ZZ_j, 22 products), and the whole
// double-and-add runs on the isomorphic curve (x, y) -> (x D_zz, y D_zzz) — for a = 0 neither the addition nor the doubling formulas contain a curve constant —
// where the table points are AFFINE: 86 MIXED additions of 10 products instead of 86 full additions of 14, table entries of 18 words instead of 36 in LDS,
// and two products at the end take the accumulator back (ZZ D_zz, ZZZ D_zzz).  129 doublings + 86 mixed additions + 43 products by beta + 56 for the table
// = ~2100 field products instead of ~2400.  The mixed addition is ec29_dev.h's in the operand-scanning forms (a lone wave per SIMD: field29_dev.h).
__device__ __forceinline__ void xyzzw_add_mixed_os(XyzzW &acc, const AffW &q, bool neg_q) {
    if (is_inf(acc)) { acc.x = q.x; acc.y = neg_q ? neg2(q.y) : q.y; acc.zz = w_one<FqW>(); acc.zzz = w_one<FqW>(); return; }
    const FqW9 u2 = LM(q.x, acc.zz), s2 = LM(q.y, acc.zzz);
    const FqW9 p = sub6(u2, acc.x);
    FqW9 r;
    {
        const uint32_t m = neg_q ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < 9; i++) r.l[i] = FqW::PAD8[i] - acc.y.l[i] + ((s2.l[i] ^ m) - m);
        r = normw(r);
    }
    if (maybe_zero_mod_p(p)) { xyzzw_add_mixed_special(acc, q, neg_q, p, r); return; }       // P == +-Q: rare
    const FqW9 pp = LS(p), rr = LS(r), ppp = LM(p, pp), qq = LM(acc.x, pp);
    FqW9 x3;
#pragma unroll
    for (int i = 0; i < 9; i++) x3.l[i] = rr.l[i] + FqW::PAD4[i] - ppp.l[i] - 2 * qq.l[i];
    x3 = normw(x3);
    const FqW9 zz3 = LM(acc.zz, pp), zzz3 = LM(acc.zzz, ppp);
    acc.y = LMA(r, sub6(qq, x3), acc.y, neg2(ppp));           // R*(Q - X3) - Y*PPP, one reduction
    acc.x = x3; acc.zz = zz3; acc.zzz = zzz3;
}
constexpr size_t G1NTT_LDS_ISO = (size_t)G1NTT_TABLE * 18 * G1NTT_THREADS * sizeof(uint32_t);   // 73728 B
__device__ __forceinline__ void lds_put_aff(uint32_t *tab, int e, const FqW9 &x, const FqW9 &y) {
#pragma unroll
    for (int k = 0; k < 9; k++) { tab[(e * 18 + k) * G1NTT_THREADS + threadIdx.x] = x.l[k]; tab[(e * 18 + 9 + k) * G1NTT_THREADS + threadIdx.x] = y.l[k]; }
}
__device__ __forceinline__ AffW lds_get_aff(const uint32_t *tab, int e) {
    AffW q;
#pragma unroll
    for (int k = 0; k < 9; k++) { q.x.l[k] = tab[(e * 18 + k) * G1NTT_THREADS + threadIdx.x]; q.y.l[k] = tab[(e * 18 + 9 + k) * G1NTT_THREADS + threadIdx.x]; }
    return q;
}
__device__ __forceinline__ XyzzW g1_mul_scalar_iso(const XyzzW &b, const Fr &k, uint32_t *tab) {
    if (is_inf(b)) return xyzzw_identity();
    FqW9 dzz, dzzz;
    {   // table: b, 2b, 3b, 4b over one denominator pair
        XyzzW t1 = b, t2 = b;
        g1_double_call(&t2);
        XyzzW t3 = t2; g1_add_call(&t3, &b);
        XyzzW t4 = t2; g1_double_call(&t4);
        {
            const FqW9 p12 = LM(t1.zz, t2.zz), p34 = LM(t3.zz, t4.zz);
            t1.x = LM(t1.x, LM(t2.zz, p34)); t2.x = LM(t2.x, LM(t1.zz, p34)); t3.x = LM(t3.x, LM(p12, t4.zz)); t4.x = LM(t4.x, LM(p12, t3.zz));
            dzz = LM(p12, p34);
        }
        {
            const FqW9 p12 = LM(t1.zzz, t2.zzz), p34 = LM(t3.zzz, t4.zzz);
            t1.y = LM(t1.y, LM(t2.zzz, p34)); t2.y = LM(t2.y, LM(t1.zzz, p34)); t3.y = LM(t3.y, LM(p12, t4.zzz)); t4.y = LM(t4.y, LM(p12, t3.zzz));
            dzzz = LM(p12, p34);
        }
        lds_put_aff(tab, 0, t1.x, t1.y); lds_put_aff(tab, 1, t2.x, t2.y); lds_put_aff(tab, 2, t3.x, t3.y); lds_put_aff(tab, 3, t4.x, t4.y);
    }
    uint32_t dig[2][6];
    uint32_t flip[2];
    {
        const GlvSplit sp = glv_split(k.l);
        glv_digits(sp.k1, dig[0]);
        glv_digits(sp.k2, dig[1]);
        flip[0] = sp.neg1 ? 8u : 0u; flip[1] = sp.neg2 ? 8u : 0u;
    }
    FqW9 beta;
#pragma unroll
    for (int i = 0; i < 9; i++) beta.l[i] = glv::BETA_W[i];
    XyzzW acc = xyzzw_identity();
    for (int w = 42; w >= 0; w--) {
        for (int r = 0; r < 3; r++) acc = xyzzw_double(acc);      // one inlined doubling site (identity passes through)
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t code = ((dig[h][w >> 3] >> (4 * (w & 7))) & 15u), mag = code & 7u;
            if (mag) {
                AffW q = lds_get_aff(tab, (int)mag - 1);
                if (h) q.x = LM(q.x, beta);                        // phi: x -> beta x (the scaling commutes with it)
                xyzzw_add_mixed_os(acc, q, ((code ^ flip[h]) & 8u) != 0);     // the one inlined mixed-addition site
            }
        }
    }
    acc.zz = LM(acc.zz, dzz); acc.zzz = LM(acc.zzz, dzzz);       // back from the isomorphic curve (the identity stays the identity)
    return acc;
}

// ---- and with EIGHT effectively affine table points b .. 8b (the table entries are half as large now: the same 147 KB of LDS hold twice as many) and signed
// 4-bit windows: 128 doublings + 64 mixed additions + 32 products by beta + ~138 for the table = ~1960 field products (2400 in rounds 4-5, 2120 with four
// entries).  The eight points' X, Y wait in the LDS slots while the prefix / suffix products of their ZZ and ZZZ are formed in registers.
constexpr int G1NTT_TABLE8 = 8;
constexpr size_t G1NTT_LDS_ISO8 = (size_t)G1NTT_TABLE8 * 18 * G1NTT_THREADS * sizeof(uint32_t);   // 147456 B
__device__ __forceinline__ XyzzW g1_mul_scalar_iso8(const XyzzW &b, const Fr &k, uint32_t *tab) {
    if (is_inf(b)) return xyzzw_identity();
    FqW9 dzz, dzzz;
    {
        FqW9 zz[8], zzz[8];
        {   // b, 2b, .. 8b: X, Y to the LDS slots, ZZ / ZZZ stay
            XyzzW t[8];
            t[0] = b;
            t[1] = b; g1_double_call(&t[1]);
            t[2] = t[1]; g1_add_call(&t[2], &b);
            t[3] = t[1]; g1_double_call(&t[3]);
            t[4] = t[3]; g1_add_call(&t[4], &b);
            t[5] = t[2]; g1_double_call(&t[5]);
            t[6] = t[5]; g1_add_call(&t[6], &b);
            t[7] = t[3]; g1_double_call(&t[7]);
#pragma unroll
            for (int i = 0; i < 8; i++) { lds_put_aff(tab, i, t[i].x, t[i].y); zz[i] = t[i].zz; zzz[i] = t[i].zzz; }
        }
        // cofactor of entry i = product of the other seven: prefix * suffix; the X (Y) of the slot is scaled in place
        auto scale = [&](FqW9 (&z)[8], int off) __attribute__((always_inline)) -> FqW9 {
            FqW9 pre[8];                                          // pre[i] = z_0 .. z_{i-1}
            pre[1] = z[0];
#pragma unroll
            for (int i = 2; i < 8; i++) pre[i] = LM(pre[i - 1], z[i - 1]);
            FqW9 suf = z[7];                                      // z_{i+1} .. z_7 while walking down
            const FqW9 all = LM(pre[7], z[7]);
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                FqW9 c;
                if (i == 7) c = pre[7]; else if (i == 0) c = suf; else c = LM(pre[i], suf);
                FqW9 v;
#pragma unroll
                for (int kk = 0; kk < 9; kk++) v.l[kk] = tab[(i * 18 + off + kk) * G1NTT_THREADS + threadIdx.x];
                v = LM(v, c);
#pragma unroll
                for (int kk = 0; kk < 9; kk++) tab[(i * 18 + off + kk) * G1NTT_THREADS + threadIdx.x] = v.l[kk];
                if (i > 0 && i < 7) suf = LM(suf, z[i]);
            }
            return all;
        };
        dzz = scale(zz, 0);
        dzzz = scale(zzz, 9);
    }
    uint32_t dig[2][6];
    uint32_t flip[2];
    {
        const GlvSplit sp = glv_split(k.l);
        glv_digits4(sp.k1, dig[0]);
        glv_digits4(sp.k2, dig[1]);
        flip[0] = sp.neg1 ? 16u : 0u; flip[1] = sp.neg2 ? 16u : 0u;
    }
    FqW9 beta;
#pragma unroll
    for (int i = 0; i < 9; i++) beta.l[i] = glv::BETA_W[i];
    XyzzW acc = xyzzw_identity();
    for (int w = 31; w >= 0; w--) {
        for (int r = 0; r < 4; r++) acc = xyzzw_double(acc);      // one inlined doubling site (identity passes through)
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const uint32_t code = (dig[h][w / 6] >> (5 * (w % 6))) & 31u, mag = code & 15u;
            if (mag) {
                AffW q = lds_get_aff(tab, (int)mag - 1);
                if (h) q.x = LM(q.x, beta);                        // phi: x -> beta x (the scaling commutes with it)
                xyzzw_add_mixed_os(acc, q, ((code ^ flip[h]) & 16u) != 0);    // the one inlined mixed-addition site
            }
        }
    }
    acc.zz = LM(acc.zz, dzz); acc.zzz = LM(acc.zzz, dzzz);       // back from the isomorphic curve (the identity stays the identity)
    return acc;
}

// pts[bitrev(i)] = in[i]   (in: affine, external form; pts: XYZZ on the 29-bit layer).  The factor 1/N is applied by the LAST stage
// (g1ntt_stage, SCALE): there half of the points are multiplied by a twiddle anyway, which takes 1/N along for nothing, so the
// scaling costs N/2 scalar multiplications instead of the N of a pass of its own (one stage's worth of the transform's 22).
__global__ void __launch_bounds__(256) g1ntt_load(XyzzW *pts, const G1Affine *in, uint32_t log_n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const G1Affine a = load_affine(in + i);
    XyzzW p = xyzzw_identity();
    if (!is_inf(a)) {
        p.x = csub_p(w_from_s(unpack<FqW>(a.x))); p.y = csub_p(w_from_s(unpack<FqW>(a.y)));
        p.zz = w_one<FqW>(); p.zzz = w_one<FqW>();
    }
    store_xyzzw(pts + brev32(i, log_n), p);
}

// one DIT stage with half-size h = 2^s; SCALE (the last stage): both outputs times n_inv (Montgomery form) — (A n_inv) +- (w n_inv) B
template <bool SCALE, int ISO>
__global__ void __launch_bounds__(G1NTT_THREADS, 1) g1ntt_stage(XyzzW *pts, uint32_t log_n, uint32_t s, PowTable tw_inv, Fr n_inv) {
    extern __shared__ uint32_t g1tab[];
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (1u << (log_n - 1))) return;
    // butterfly j of the stage: group jh, position jl inside the half (twiddle exponent).  jl = 0 needs no multiplication; in the early
    // stages (h < 64) consecutive lanes would all differ in jl and every wave would pay for the one lane in h that has it.  There the
    // position is the SLOW index of the launch: whole waves (workgroups) share jl, and those with jl = 0 — half of stage 1, a quarter of
    // stage 2, .. — only add.  (The accesses become strided; memory traffic is nothing in this kernel.)
    const uint32_t h = 1u << s;
    uint32_t jl, jh;
    if (s < 6 && log_n >= 8) { jl = j >> (log_n - 1 - s); jh = j & ((1u << (log_n - 1 - s)) - 1); }
    else { jl = j & (h - 1); jh = j >> s; }
    const uint32_t i0 = (jh << (s + 1)) | jl, i1 = i0 + h;
    XyzzW a = load_xyzzw(pts + i0), b = load_xyzzw(pts + i1);
    if (jl || SCALE) {
        Fr w = n_inv;
        if (jl) {
            // omega_N^-(jl * N / 2h)
            uint32_t e = (jl << (log_n - s - 1)) << (MAX_LOG_N - log_n);
            w = mul(load_fp(tw_inv.lo + (e & (POW_TAB - 1))), load_fp(tw_inv.hi + (e >> POW_SPLIT)));
            if (SCALE) w = mul(w, n_inv);
        }
        auto mul = [&](const XyzzW &pt, const Fr &kk) __attribute__((always_inline)) {
            if constexpr (ISO == 2) return g1_mul_scalar_iso8(pt, kk, g1tab);
            else if constexpr (ISO == 1) return g1_mul_scalar_iso(pt, kk, g1tab);
            else return g1_mul_scalar(pt, kk, g1tab);
        };
        b = mul(b, to_canonical(w));
        if (SCALE) a = mul(a, to_canonical(n_inv));
    }
    XyzzW lo = a, nb = b;
    nb.y = sub6(w_zero<FqW>(), b.y);
    g1_add_call(&lo, &b);
    g1_add_call(&a, &nb);
    store_xyzzw(pts + i0, lo);
    store_xyzzw(pts + i1, a);
}

// XYZZ -> affine with ONE field inversion per G1NTT_NORM_K points (Montgomery's trick on ZZZ; 1 / Z = ZZ / ZZZ, as srs_normalise_kernel does for the MSM table):
// a Fermat inversion per point was 4.3 ms of a 2^20-point transform.  A thread takes K consecutive points; the second walk reads them again (four products each)
// rather than keep K x 32 words alive.  Same canonical coordinates as before, bit for bit.
constexpr uint32_t G1NTT_NORM_K = 8;
__global__ void __launch_bounds__(256) g1ntt_to_affine(G1Affine *out, const XyzzW *pts, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lo = t * G1NTT_NORM_K, hi = lo + G1NTT_NORM_K < n ? lo + G1NTT_NORM_K : n;
    if (lo >= n) return;
    Fq prefix[G1NTT_NORM_K];
    Fq acc = Fq::one();
#pragma unroll
    for (uint32_t j = 0; j < G1NTT_NORM_K; j++) {
        if (lo + j < hi) {
            const XyzzW w = load_xyzzw(pts + lo + j);
            if (!is_inf(w)) acc = mul(acc, pack<FqParams>(s_from_w(w.zzz)));
        }
        prefix[j] = acc;
    }
    Fq inv_acc = inv(acc);
#pragma unroll
    for (uint32_t jj = 0; jj < G1NTT_NORM_K; jj++) {
        const uint32_t j = G1NTT_NORM_K - 1 - jj;
        if (lo + j >= hi) continue;
        const G1Xyzz q = xyzzw_export(load_xyzzw(pts + lo + j));  // back to the external form (canonical, R = 2^256); the identity is all zero
        G1Affine a; a.x = Fq::zero(); a.y = Fq::zero();
        if (!is_inf(q)) {
            const Fq zi = j ? mul(inv_acc, prefix[j - 1]) : inv_acc;  // 1 / ZZZ_j
            inv_acc = mul(inv_acc, q.zzz);
            const Fq iz = mul(q.zz, zi), izz = mul(iz, iz);       // 1 / Z, 1 / ZZ
            a.x = mul(q.x, izz);
            a.y = mul(q.y, zi);
        }
        store_fp(&out[lo + j].x, a.x);
        store_fp(&out[lo + j].y, a.y);
    }
}

int32_t g1_intt_dev(plk_ctx *ctx, const G1Affine *in, uint32_t log_n, G1Affine *out, hipStream_t st) {
    if (log_n > 26) { set_error("g1_intt: size exceeds 2^26"); return PLK_ERR_SIZE; }
    PLK_TRY(ntt_init_tables(ctx));
    const uint32_t n = 1u << log_n;
    // the transform borrows the scratch of the first commitment slot: nothing may be in flight there
    if (ctx->msm_enq != ctx->msm_fin) { set_error("g1_intt: a commitment enqueued with plk_msm_g1_enqueue_dev is still in flight (call plk_msm_g1_finish first)"); return PLK_ERR_ARG; }
    PLK_TRY(ctx->slot[0].c.reserve((size_t)n * sizeof(XyzzW)));
    XyzzW *pts = ctx->slot[0].c.as<XyzzW>();
    const Fr n_inv = ctx->n_inv[log_n];                           // Montgomery form; 1 for log_n = 0 (no stage, nothing to scale)
    static std::atomic<bool> attr_set{false};
    if (!attr_set) {
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS_ISO));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS_ISO));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS_ISO8));
        PLK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g1ntt_stage<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G1NTT_LDS_ISO8));
        attr_set = true;
    }
    // PLK_G1NTT_ISO (A/B knob): 0 = the window table in XYZZ and full additions, as in rounds 4-5; 1 = four effectively affine entries, 3-bit windows; default 2 = eight, 4-bit windows
    // (the same chain in Jacobian coordinates — 4S + 3M doublings — was built, verified and measured 3 % SLOWER: profiles/r06b_g1ntt_jacobian.patch)
    static const int iso = [] { const char *e = getenv("PLK_G1NTT_ISO"); return e ? atoi(e) : 2; }();
    hipLaunchKernelGGL(g1ntt_load, dim3((n + 255) / 256), dim3(256), 0, st, pts, in, log_n);
    const dim3 sgrid((n / 2 + G1NTT_THREADS - 1) / G1NTT_THREADS);
    auto stage = [&](auto scale_tag, uint32_t s) {
        constexpr bool SC = decltype(scale_tag)::value;
        if (iso == 2) hipLaunchKernelGGL((g1ntt_stage<SC, 2>), sgrid, dim3(G1NTT_THREADS), G1NTT_LDS_ISO8, st, pts, log_n, s, ctx->tw_inv, n_inv);
        else if (iso == 1) hipLaunchKernelGGL((g1ntt_stage<SC, 1>), sgrid, dim3(G1NTT_THREADS), G1NTT_LDS_ISO, st, pts, log_n, s, ctx->tw_inv, n_inv);
        else hipLaunchKernelGGL((g1ntt_stage<SC, 0>), sgrid, dim3(G1NTT_THREADS), G1NTT_LDS, st, pts, log_n, s, ctx->tw_inv, n_inv);
    };
    for (uint32_t s = 0; s + 1 < log_n; s++) stage(std::false_type{}, s);
    if (log_n) stage(std::true_type{}, log_n - 1);
    hipLaunchKernelGGL(g1ntt_to_affine, dim3(((n + G1NTT_NORM_K - 1) / G1NTT_NORM_K + 255) / 256), dim3(256), 0, st, out, (const XyzzW *)pts, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk

using namespace plk;

extern "C" int32_t plk_g1_intt(plk_ctx *ctx, const plk_g1_affine *in, uint32_t log_n, plk_g1_affine *out) {
    if (!ctx || !in || !out) { set_error("plk_g1_intt: bad argument"); return PLK_ERR_ARG; }
    if (log_n > 26) { set_error("g1_intt: size exceeds 2^26"); return PLK_ERR_SIZE; }
    PLK_HIP(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(plk_g1_affine) << log_n;
    PLK_TRY(ctx->stage.reserve(2 * bytes));
    G1Affine *d_in = ctx->stage.as<G1Affine>(), *d_out = d_in + ((size_t)1 << log_n);
    PLK_HIP(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLK_TRY(g1_intt_dev(ctx, d_in, log_n, d_out, ctx->stream));
    PLK_HIP(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

// the same transform applied to the first 2^log_n points of the resident SRS; result left on the device
extern "C" int32_t plk_g1_intt_srs_dev(plk_ctx *ctx, uint32_t log_n, void *out_dev, void *stream) {
    if (!ctx || !out_dev) { set_error("plk_g1_intt_srs_dev: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->srs || ctx->srs_n < ((uint64_t)1 << log_n)) { set_error("g1_intt: SRS too small"); return PLK_ERR_SRS; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return g1_intt_dev(ctx, reinterpret_cast<const G1Affine *>(ctx->srs), log_n, reinterpret_cast<G1Affine *>(out_dev), s);
}
