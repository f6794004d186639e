// Host-side check of the 9 x 29-bit lazy field layer and its group law (field29_dev.h / ec29_dev.h: the PLK_HD functions
// compile for the host too) against the 8 x 32-bit layer (field_dev.h / ec_dev.h), which the GPU tests pin to the oracle.
// Built and run by tests/test_field29_host.py with hipcc; needs no GPU.
#include "ec29_dev.h"
#include "glv_dev.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
using namespace plk;
static uint64_t rs = 88172645463325252ULL;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
template <class PR> Fp<PR> rnd_fp() { Fp<PR> a; for (int i = 0; i < 8; i++) a.l[i] = rnd(); a.l[7] &= 0x0fffffff; return a; }   // < 2^252 < p
static G1Affine to_aff(const G1Xyzz &p) {
    G1Affine a; if (is_inf(p)) { a.x = Fq::zero(); a.y = Fq::zero(); return a; }
    Fq i = inv(mul(p.zz, p.zzz)); a.x = mul(p.x, mul(i, p.zzz)); a.y = mul(p.y, mul(i, p.zz)); return a; }
static AffW aff_to_w(const G1Affine &a) { AffW r; r.x = csub_p(w_from_s(unpack<FqW>(a.x))); r.y = csub_p(w_from_s(unpack<FqW>(a.y))); return r; }
static G1Xyzz exportw(const XyzzW &p) {
    if (is_inf(p)) return xyzz_identity();
    G1Xyzz r; r.x = pack<FqParams>(s_from_w(p.x)); r.y = pack<FqParams>(s_from_w(p.y)); r.zz = pack<FqParams>(s_from_w(p.zz)); r.zzz = pack<FqParams>(s_from_w(p.zzz)); return r; }
int main() {
    int bad = 0;
    for (int it = 0; it < 20000; it++) {
        Fr a = rnd_fp<FrParams>(), b = rnd_fp<FrParams>();
        FrW9 aw = w_from_s(unpack<FrW>(a)), bw = w_from_s(unpack<FrW>(b));
        Fr got = pack<FrParams>(s_from_w(mulw(aw, bw)));
        if (got != mul(a, b)) { bad++; if (bad < 5) printf("mul mismatch %d\n", it); }
        // linear use: raw 256-domain data times a W-domain constant
        Fr got2 = pack<FrParams>(csub_p(mulw(unpack<FrW>(a), bw)));
        if (got2 != mul(a, b)) { bad++; if (bad < 5) printf("linear mul mismatch %d\n", it); }
        // lazy add/sub chains
        FrW9 s = addn(addn(aw, bw), aw);                 // 2a + b
        FrW9 d = sub2(s, bw);                            // 2a (+2p)
        FrW9 d2 = sub6(d, addn(addn(aw, aw), aw));       // -a (+6p)
        Fr want = neg(a);
        if (pack<FrParams>(s_from_w(d2)) != want) { bad++; if (bad < 5) printf("addsub mismatch %d\n", it); }
        if (!is_zero_mod_p(sub4(addn(aw, bw), addn(bw, aw)))) { bad++; if (bad < 5) printf("zero test %d\n", it); }
        if (is_zero_mod_p(sub4(addn(aw, bw), addn(bw, bw))) && a != b) { bad++; if (bad < 5) printf("zero test false positive %d\n", it); }
    }
    // dedicated squaring, fused products, cheap canonical reduction — both fields
    for (int it = 0; it < 20000; it++) {
        Fq a = rnd_fp<FqParams>(), b = rnd_fp<FqParams>(), c = rnd_fp<FqParams>(), d = rnd_fp<FqParams>(), e = rnd_fp<FqParams>(), f = rnd_fp<FqParams>();
        FqW9 aw = w_from_s(unpack<FqW>(a)), bw = w_from_s(unpack<FqW>(b)), cw = w_from_s(unpack<FqW>(c)), dw = w_from_s(unpack<FqW>(d)),
             ew = w_from_s(unpack<FqW>(e)), fw = w_from_s(unpack<FqW>(f));
        if (pack<FqParams>(s_from_w(sqrw(aw))) != mul(a, a)) { bad++; if (bad < 5) printf("sqrw mismatch %d\n", it); }
        if (pack<FqParams>(s_from_w(mul2addw(aw, bw, cw, dw))) != add(mul(a, b), mul(c, d))) { bad++; if (bad < 5) printf("mul2addw mismatch %d\n", it); }
        if (pack<FqParams>(s_from_w(mulsum3w(aw, bw, cw, dw, ew, fw))) != add(add(mul(a, b), mul(c, d)), mul(e, f))) { bad++; if (bad < 5) printf("mulsum3w mismatch %d\n", it); }
        FqW9 big = addn(addn(addn(aw, bw), addn(cw, dw)), addn(ew, fw));             // < 6.6p, normalised
        for (int k = 0; k < (it & 3); k++) big = addn(big, big);                      // up to < 53p
        FqW9 r1 = reduce_small(big), r2 = reduce_full(big);                           // same residue (x * one / R)
        for (int k = 0; k < 9; k++) if (r1.l[k] != r2.l[k]) { bad++; if (bad < 5) printf("reduce_small mismatch %d\n", it); break; }
    }
    // the operand-scanning forms (latency-bound EC chains) and the two-chain lockstep forms are the SAME functions, bit for bit
    for (int it = 0; it < 20000; it++) {
        Fq a = rnd_fp<FqParams>(), b = rnd_fp<FqParams>(), c = rnd_fp<FqParams>(), d = rnd_fp<FqParams>();
        FqW9 aw = w_from_s(unpack<FqW>(a)), bw = w_from_s(unpack<FqW>(b)), cw = w_from_s(unpack<FqW>(c)), dw = w_from_s(unpack<FqW>(d));
        if (it & 1) aw = addw(aw, sub2(cw, dw));                                          // an un-normalised left operand now and then
        const FqW9 m1 = mulw(aw, bw), m2 = mulw_os(aw, bw), s1 = sqrw(bw), s2 = sqrw_os(bw);
        const FqW9 f1 = mul2addw(normw(aw), bw, cw, dw), f2 = mul2addw_os(normw(aw), bw, cw, dw);
        FqW9 p0, p1, q0, q1;
        mulw2(aw, bw, cw, dw, p0, p1);
        sqrw2(bw, dw, q0, q1);
        const FqW9 p1s = mulw(cw, dw), q1s = sqrw(dw);
        for (int k = 0; k < 9; k++)
            if (m1.l[k] != m2.l[k] || s1.l[k] != s2.l[k] || f1.l[k] != f2.l[k] || p0.l[k] != m1.l[k] || p1.l[k] != p1s.l[k] || q0.l[k] != s1.l[k] || q1.l[k] != q1s.l[k]) {
                bad++; if (bad < 5) printf("product forms disagree %d\n", it); break; }
    }
    // product by a constant held as three shifted copies (mul_tw3: 108 multiply-adds, the NTT's stage twiddles): equals the plain
    // product for normalised and for un-normalised operands up to the limb bound, result < 4p for normalised ones, and the lockstep
    // pair is the same function
    {
        FrW9 four_p = w_zero<FrW>();
        { uint64_t c = 0; for (int i = 0; i < 9; i++) { c += 4ull * FrW::P29[i]; four_p.l[i] = (uint32_t)c & M29; c >>= 29; } }
        auto less = [](const FrW9 &a, const FrW9 &b) { for (int i = 8; i >= 0; i--) if (a.l[i] != b.l[i]) return a.l[i] < b.l[i]; return false; };
        for (int it = 0; it < 20000; it++) {
            Fr a = rnd_fp<FrParams>(), b = rnd_fp<FrParams>(), t = rnd_fp<FrParams>();
            const Tw3<FrW> tw = make_tw3(csub_p(w_from_s(unpack<FrW>(t))));
            FrW9 x = unpack<FrW>(a), y = unpack<FrW>(b);
            if ((it & 3) == 1) { for (int k = 0; k < 9; k++) x.l[k] = M29; }                  // largest normalised operand (2^261 - 1)
            if ((it & 3) == 2) { for (int k = 0; k < 8; k++) x.l[k] += FrW::PAD4[k]; }        // un-normalised, like a padded difference
            if ((it & 3) == 3) { for (int k = 0; k < 8; k++) x.l[k] = MULTW3_X_LIMB_MAX; x.l[8] = 0x003fffffu; }
            const FrW9 r = mul_tw3(x, tw);
            if ((it & 3) < 2 && !less(r, four_p)) { bad++; if (bad < 5) printf("mul_tw3 result not below 4p at %d\n", it); }
            for (int k = 0; k < 8; k++) if (r.l[k] > M29) { bad++; if (bad < 5) printf("mul_tw3 output limb not normalised %d\n", it); break; }
            // reference: the ordinary product of the normalised operand by the same constant in the W domain
            const FrW9 want = reduce_small(mulw(normw(x), csub_p(w_from_s(unpack<FrW>(t))))), got = reduce_small(r);
            for (int k = 0; k < 9; k++) if (want.l[k] != got.l[k]) { bad++; if (bad < 5) printf("mul_tw3 mismatch %d (case %d)\n", it, it & 3); break; }
            if ((it & 3) == 0 && pack<FrParams>(got) != mul(a, t)) { bad++; if (bad < 5) printf("mul_tw3 vs Fr mul mismatch %d\n", it); }
            FrW9 r0, r1;
            mul_tw3_2(x, y, tw, r0, r1);
            const FrW9 r1s = mul_tw3(y, tw);
            for (int k = 0; k < 9; k++) if (r0.l[k] != r.l[k] || r1.l[k] != r1s.l[k]) { bad++; if (bad < 5) printf("mul_tw3_2 disagrees %d\n", it); break; }
        }
    }
    // mulw with an un-normalised left operand (ntt.hip r4_finish): limbs at MULW_A_LIMB_MAX against a right operand with
    // every limb at 2^29 - 1, raw padded differences against their normalised forms, and one lazy radix-4 butterfly
    // against the strictly normalised formulas
    {
        FrW9 amax, bmax;
        for (int i = 0; i < 9; i++) { amax.l[i] = MULW_A_LIMB_MAX; bmax.l[i] = M29; }
        amax.l[8] = 0x00ffffffu; bmax.l[8] = 0x00ffffffu;                              // keep the values inside the 2^261 capacity
        FrW9 r1 = s_from_w(mulw(amax, bmax)), r2 = s_from_w(mulw(normw(amax), bmax));
        for (int k = 0; k < 9; k++) if (r1.l[k] != r2.l[k]) { bad++; printf("mulw at the limb bound\n"); break; }
    }
    for (int it = 0; it < 20000; it++) {
        Fr a = rnd_fp<FrParams>(), b = rnd_fp<FrParams>(), c = rnd_fp<FrParams>(), d = rnd_fp<FrParams>(), t = rnd_fp<FrParams>(), t2 = rnd_fp<FrParams>(), t3 = rnd_fp<FrParams>();
        FrW9 x0 = unpack<FrW>(a), x1 = unpack<FrW>(b), x2 = unpack<FrW>(c), x3 = unpack<FrW>(d);
        FrW9 w1 = w_from_s(unpack<FrW>(t)), w2 = w_from_s(unpack<FrW>(t2)), w3 = w_from_s(unpack<FrW>(t3));
        if (it & 1) for (int k = 0; k < 9; k++) { x0.l[k] = M29; x2.l[k] = M29; }       // largest normalised limbs
        x0.l[8] &= 0x00ffffffu; x2.l[8] &= 0x00ffffffu;
        // strict
        FrW9 y1 = mulw(x1, w1), y3 = mulw(x3, w1);
        FrW9 b0 = addn(x0, y1), b1 = sub2(x0, y1), b2 = mulw(addn(x2, y3), w2), b3 = mulw(sub2(x2, y3), w3);
        FrW9 s0 = addn(b0, b2), s2 = sub2(b0, b2), s1 = addn(b1, b3), s3 = sub2(b1, b3);
        // lazy
        FrW9 u, v, o0, o1, o2, o3;
        for (int i = 0; i < 9; i++) { u.l[i] = x2.l[i] + y3.l[i]; v.l[i] = x2.l[i] + FrW::PAD2[i] - y3.l[i]; }
        for (int i = 0; i < 9; i++) if (v.l[i] > MULW_A_LIMB_MAX || u.l[i] > MULW_A_LIMB_MAX) { bad++; printf("limb bound exceeded\n"); break; }
        FrW9 c2 = mulw(u, w2), c3 = mulw(v, w3);
        for (int i = 0; i < 9; i++) {
            const uint32_t bb = x0.l[i] + y1.l[i];
            o0.l[i] = bb + c2.l[i]; o2.l[i] = bb + FrW::PAD4[i] - c2.l[i];
            o1.l[i] = x0.l[i] + FrW::PAD2[i] - y1.l[i] + c3.l[i]; o3.l[i] = x0.l[i] + FrW::PAD4[i] - y1.l[i] - c3.l[i];
        }
        FrW9 lazy[4] = {normw(o0), normw(o1), normw(o2), normw(o3)}, strict[4] = {s0, s1, s2, s3};
        for (int q = 0; q < 4; q++) {
            FrW9 r1 = s_from_w(lazy[q]), r2 = s_from_w(strict[q]);
            for (int k = 0; k < 9; k++) if (r1.l[k] != r2.l[k]) { bad++; if (bad < 5) printf("lazy butterfly output %d mismatch at %d\n", q, it); break; }
        }
    }
    printf("field29: %d mismatches\n", bad);
    // EC: random chain of mixed adds, doubles and full adds against ec_dev.h
    G1Affine g; g.x = from_u64<FqParams>(1); g.y = from_u64<FqParams>(2);
    G1Xyzz acc = xyzz_identity(); XyzzW accw = xyzzw_identity();
    G1Affine pts[8]; AffW ptsw[8];
    { G1Xyzz t = xyzz_from_affine(g); for (int i = 0; i < 8; i++) { pts[i] = to_aff(t); ptsw[i] = aff_to_w(pts[i]); t = xyzz_double(t); xyzz_add_mixed(t, g, false); } }
    int ebad = 0;
    for (int it = 0; it < 3000; it++) {
        int k = rnd() & 7; bool ng = rnd() & 1; int op = rnd() % 10;
        if (op < 7) { xyzz_add_mixed(acc, pts[k], ng); xyzzw_add_mixed(accw, ptsw[k], ng); }
        else if (op == 7) { acc = xyzz_double(acc); accw = xyzzw_double(accw); }
        else if (op == 8) { G1Xyzz o = acc; xyzz_add_mixed(o, pts[k], !ng); xyzz_add(acc, o); XyzzW ow = accw; xyzzw_add_mixed(ow, ptsw[k], !ng); xyzzw_add(accw, ow); }
        else { /* P + P and P - P through the mixed add */ G1Affine cur = to_aff(acc); AffW cw = aff_to_w(cur); bool n2 = rnd() & 1; xyzz_add_mixed(acc, cur, n2); xyzzw_add_mixed(accw, cw, n2); }
        G1Affine x = to_aff(acc), y = to_aff(exportw(accw));
        if (x.x != y.x || x.y != y.y) { ebad++; if (ebad < 5) printf("ec mismatch at %d op %d\n", it, op); }
    }
    printf("ec29: %d mismatches\n", ebad);
    // known-answer lines for the Python side (checked there with big integers): canonical a, b, a*b through the W layer
    auto hex = [](const uint32_t *l) { for (int i = 7; i >= 0; i--) printf("%08x", l[i]); };
    for (int it = 0; it < 40; it++) {
        Fr a = rnd_fp<FrParams>(), b = rnd_fp<FrParams>();
        Fr p = pack<FrParams>(s_from_w(mulw(w_from_s(unpack<FrW>(a)), w_from_s(unpack<FrW>(b)))));
        Fr ac = to_canonical(a), bc = to_canonical(b), pc = to_canonical(p);
        printf("KAT Fr "); hex(ac.l); printf(" "); hex(bc.l); printf(" "); hex(pc.l); printf("\n");
        Fq c = rnd_fp<FqParams>(), d = rnd_fp<FqParams>();
        Fq q = pack<FqParams>(s_from_w(sqrw(w_from_s(unpack<FqW>(c)))));
        Fq cc = to_canonical(c), qc = to_canonical(q);
        (void)d;
        printf("KAT Fq "); hex(cc.l); printf(" "); hex(cc.l); printf(" "); hex(qc.l); printf("\n");
    }
    // GLV split of canonical scalars (glv_dev.h, used by the G1 iNTT): known-answer lines "GLV k |k1| neg1 |k2| neg2"
    // (Python checks k1 + k2 lambda = k mod r and both halves < 2^128), the signed 3-bit and 4-bit window digits re-summed here, and
    // phi(P) = (beta x, y) = lambda P on the 29-bit layer by double-and-add with lambda
    int gbad = 0;
    auto hex5 = [](const uint32_t *l) { for (int i = 4; i >= 0; i--) printf("%08x", l[i]); };
    for (int it = 0; it < 2000; it++) {
        Fr k = to_canonical(rnd_fp<FrParams>());
        if (it == 0) { for (int i = 0; i < 8; i++) k.l[i] = 0; }
        if (it == 1) { for (int i = 0; i < 8; i++) k.l[i] = FrParams::P[i]; k.l[0] -= 1; }            // r - 1
        if (it == 2) { for (int i = 0; i < 8; i++) k.l[i] = 0; k.l[0] = 1; }
        const GlvSplit sp = glv_split(k.l);
        if (sp.k1[4] != 0 || sp.k2[4] != 0) { gbad++; if (gbad < 5) printf("glv half >= 2^128 at %d\n", it); }
        for (int h = 0; h < 2; h++) {                              // digits: sum d_w 8^w must give the magnitude back
            const uint32_t *m = h ? sp.k2 : sp.k1;
            uint32_t dig[6]; glv_digits(m, dig);
            __int128 acc = 0; unsigned __int128 want = 0;
            for (int w = 42; w >= 0; w--) { const uint32_t c = (dig[w >> 3] >> (4 * (w & 7))) & 15u; const int d = (c & 8u) ? -(int)(c & 7u) : (int)(c & 7u); if ((c & 7u) > 4 || (c == 12u)) gbad++; acc = acc * 8 + d; }
            for (int i = 3; i >= 0; i--) want = (want << 32) | m[i];
            if (acc < 0 || (unsigned __int128)acc != want) { gbad++; if (gbad < 5) printf("glv digits mismatch at %d half %d\n", it, h); }
            // the signed 4-bit windows of the eight-entry table (digits in [-7, 8], 32 windows: both halves are below 2^127)
            if (m[3] >> 31) { gbad++; if (gbad < 5) printf("glv half >= 2^127 at %d\n", it); }
            uint32_t dig4[6]; glv_digits4(m, dig4);
            __int128 acc4 = 0;
            for (int w = 31; w >= 0; w--) { const uint32_t c = (dig4[w / 6] >> (5 * (w % 6))) & 31u; const int d = (c & 16u) ? -(int)(c & 15u) : (int)(c & 15u); if ((c & 15u) > 8 || c == 24u) gbad++; acc4 = acc4 * 16 + d; }
            if (acc4 < 0 || (unsigned __int128)acc4 != want) { gbad++; if (gbad < 5) printf("glv 4-bit digits mismatch at %d half %d\n", it, h); }
        }
        if (it < 40) { printf("GLV "); hex(k.l); printf(" "); hex5(sp.k1); printf(" %d ", sp.neg1 ? 1 : 0); hex5(sp.k2); printf(" %d\n", sp.neg2 ? 1 : 0); }
    }
    {
        // lambda * P by double-and-add on the W layer against (beta x, y)
        const uint32_t lam[8] = {0x36636f23u, 0xb8ca0b2du, 0xec2bc5e9u, 0xcc37a73fu, 0x3fd84104u, 0x048b6e19u, 0xe131a029u, 0x30644e72u};
        FqW9 beta; for (int i = 0; i < 9; i++) beta.l[i] = glv::BETA_W[i];
        for (int k = 0; k < 8; k++) {
            XyzzW p; p.x = ptsw[k].x; p.y = ptsw[k].y; p.zz = w_one<FqW>(); p.zzz = w_one<FqW>();
            XyzzW acc2 = xyzzw_identity();
            for (int bit = 253; bit >= 0; bit--) { acc2 = xyzzw_double(acc2); if ((lam[bit >> 5] >> (bit & 31)) & 1u) xyzzw_add(acc2, p); }
            XyzzW ph = p; ph.x = WM(p.x, beta);
            G1Affine x = to_aff(exportw(acc2)), y = to_aff(exportw(ph));
            if (x.x != y.x || x.y != y.y) { gbad++; printf("phi(P) != lambda P for point %d\n", k); }
        }
    }
    printf("glv: %d mismatches\n", gbad);
    return bad + ebad + gbad;
}
