// BN254 optimal-ate pairing on the host (see pairing.h).  Algorithm: Miller loop over the bits of
// 6u+2 = 29793968203157093288 with affine arithmetic on the twist (slopes in Fq2), two Frobenius
// correction lines, and a plain square-and-multiply final exponentiation by (p^12 - 1)/r.
#include "pairing.h"

namespace plk {
namespace host {

Fq12 Fq12::zero() { Fq12 r; for (auto &x : r.c) x = HFq::zero(); return r; }
Fq12 Fq12::one() { Fq12 r = zero(); r.c[0] = HFq::one(); return r; }
bool Fq12::is_one() const {
    if (!(c[0] == HFq::one())) return false;
    for (int i = 1; i < 12; i++) if (!c[i].is_zero()) return false;
    return true;
}

// schoolbook product, then w^k = 18 w^(k-6) - 82 w^(k-12) from the top down
Fq12 Fq12::operator*(const Fq12 &o) const {
    static const HFq k18 = HFq::from_u64(18), k82 = HFq::from_u64(82);
    HFq t[23];
    for (auto &x : t) x = HFq::zero();
    for (int i = 0; i < 12; i++) {
        if (c[i].is_zero()) continue;
        for (int j = 0; j < 12; j++) {
            if (o.c[j].is_zero()) continue;
            t[i + j] = t[i + j] + c[i] * o.c[j];
        }
    }
    for (int k = 22; k >= 12; k--) {
        if (t[k].is_zero()) continue;
        t[k - 6] = t[k - 6] + t[k] * k18;
        t[k - 12] = t[k - 12] - t[k] * k82;
    }
    Fq12 r;
    for (int i = 0; i < 12; i++) r.c[i] = t[i];
    return r;
}

static const uint64_t ATE_LOOP_HI = 1;                               // 6u + 2 = 2^64 + ATE_LOOP_LO
static const uint64_t ATE_LOOP_LO = 11347224129447541672ULL;    // 29793968203157093288 - 2^64
static const uint64_t FROB_E3[4] = {0x69602eb24829a9c2ULL, 0xdd2b2385cd7b4384ULL, 0xe81ac1e7808072c9ULL, 0x10216f7ba065e00dULL};          // (p - 1) / 3
static const uint64_t FROB_E2[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};          // (p - 1) / 2
static const int FINAL_EXP_LIMBS = 44;
static const uint64_t FINAL_EXP[FINAL_EXP_LIMBS] = {                  // (p^12 - 1) / r, 2790 bits
    0x86964b64ca86f120ULL, 0x40a4efb7e54523a4ULL, 0x837fa97896e84abbULL, 0x361102b6b9b2b918ULL,
    0xc0de81def35692daULL, 0xbe04c7e8a6c3c760ULL, 0xd766f9c9d570bb7fULL, 0xc230974d83561841ULL,
    0x5bba1668c3be69a3ULL, 0x7f3811c410526294ULL, 0x29baee7ddadda71cULL, 0xbf813b8d145da900ULL,
    0x641bbadf423f9a2cULL, 0xa80bb4ea44eacc5eULL, 0xcd65664814fde37cULL, 0x4a0364b9580291d2ULL,
    0xee93dfb10826f0ddULL, 0x6b42db8dc5514724ULL, 0xbb10cf430b0f3785ULL, 0x40494e406f804216ULL,
    0x55cfe107acf3aafbULL, 0x2088ec80e0ebae87ULL, 0x846a3ed011a337a0ULL, 0x48a45a4a1e3a5195ULL,
    0xe5664568dfc50e16ULL, 0xab6a41294c0cc4ebULL, 0x82d0d602d268c7daULL, 0x6668449aed3cc48aULL,
    0x5062cd0fb2015dfcULL, 0x7f2940a8b1ddb3d1ULL, 0x77f5b63a2a226448ULL, 0xfef0781361e443aeULL,
    0xf977870e88d5c6c8ULL, 0x790364a61f676baaULL, 0x5887e72eceaddea3ULL, 0x1377e563a09a1b70ULL,
    0x0c54efee1bd8c3b2ULL, 0x3ec3d15ad524d8f7ULL, 0xdaf15466b2383a5dULL, 0xe1e30a73bb94fec0ULL,
    0x6a1c71015f3f7be2ULL, 0x842d43bf6369b1ffULL, 0x20fddadf107d20bcULL, 0x0000002f4b6dc970ULL};

static Fq2 fq2_pow(const Fq2 &a, const uint64_t e[4]) {
    Fq2 acc = Fq2::one(), b = a;
    for (int i = 0; i < 256; i++) { if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * b; b = b.sqr(); }
    return acc;
}

static const Fq2 &xi() { static const Fq2 v = {HFq::from_u64(9), HFq::one()}; return v; }
static const Fq2 &twist_b() { static const Fq2 v = xi().inv().scale(HFq::from_u64(3)); return v; }
static const Fq2 &gamma2() { static const Fq2 v = fq2_pow(xi(), FROB_E3); return v; }   // w^(2(p-1))
static const Fq2 &gamma3() { static const Fq2 v = fq2_pow(xi(), FROB_E2); return v; }   // w^(3(p-1))

bool g2_from_bytes(const uint8_t in[128], G2Affine *out) {
    bool all_zero_tail = true;
    for (int i = 1; i < 128; i++) if (in[i]) all_zero_tail = false;
    if ((in[0] & 0x40) && all_zero_tail) { out->inf = true; out->x = Fq2::zero(); out->y = Fq2::zero(); return true; }
    out->inf = false;
    if (!HFq::from_be_bytes(in, &out->x.c1) || !HFq::from_be_bytes(in + 32, &out->x.c0)) return false;
    if (!HFq::from_be_bytes(in + 64, &out->y.c1) || !HFq::from_be_bytes(in + 96, &out->y.c0)) return false;
    return out->y.sqr() == out->x.sqr() * out->x + twist_b();
}

// line through the untwisted images of T (and U, or the tangent at T) evaluated at P = (xp, yp) in G1:
//   l = -yp + (m xp) w + (y_T - m x_T) w^3 ,  m = the slope on the twist; a + b i embeds as (a - 9b) + b w^6
static Fq12 line_value(const Fq2 &m, const Fq2 &xt, const Fq2 &yt, const HFq &xp, const HFq &yp) {
    static const HFq nine = HFq::from_u64(9);
    Fq12 l = Fq12::zero();
    l.c[0] = -yp;
    const Fq2 a = m.scale(xp), b = yt - m * xt;
    l.c[1] = a.c0 - a.c1 * nine;  l.c[7] = a.c1;
    l.c[3] = b.c0 - b.c1 * nine;  l.c[9] = b.c1;
    return l;
}

struct TwistPoint { Fq2 x, y; };

// f *= l_{R,R}(P);  R = 2R
static void step_double(Fq12 &f, TwistPoint &R, const HFq &xp, const HFq &yp) {
    const Fq2 x2 = R.x.sqr();
    const Fq2 m = (x2 + x2 + x2) * (R.y + R.y).inv();
    f = f * line_value(m, R.x, R.y, xp, yp);
    const Fq2 x3 = m.sqr() - R.x - R.x;
    R.y = m * (R.x - x3) - R.y;
    R.x = x3;
}
// f *= l_{R,Q}(P);  R = R + Q   (R != +-Q for points of prime order r inside the loop)
static void step_add(Fq12 &f, TwistPoint &R, const TwistPoint &Q, const HFq &xp, const HFq &yp) {
    const Fq2 m = (Q.y - R.y) * (Q.x - R.x).inv();
    f = f * line_value(m, R.x, R.y, xp, yp);
    const Fq2 x3 = m.sqr() - R.x - Q.x;
    R.y = m * (R.x - x3) - R.y;
    R.x = x3;
}

static Fq12 miller_loop(const HAffine &P, const G2Affine &Qin) {
    Fq12 f = Fq12::one();
    if (P.is_inf() || Qin.inf) return f;
    const TwistPoint Q = {Qin.x, Qin.y};
    TwistPoint R = Q;
    for (int i = 63; i >= 0; i--) {
        f = f * f;
        step_double(f, R, P.x, P.y);
        if ((ATE_LOOP_LO >> i) & 1) step_add(f, R, Q, P.x, P.y);
    }
    (void)ATE_LOOP_HI;
    const TwistPoint Q1 = {Q.x.conj() * gamma2(), Q.y.conj() * gamma3()};              // pi(Q)
    const TwistPoint Q2n = {Q1.x.conj() * gamma2(), -(Q1.y.conj() * gamma3())};        // -pi^2(Q)
    step_add(f, R, Q1, P.x, P.y);
    step_add(f, R, Q2n, P.x, P.y);
    return f;
}

bool pairing_product_is_one(const HAffine *g1, const G2Affine *g2, int pairs) {
    Fq12 f = Fq12::one();
    for (int i = 0; i < pairs; i++) f = f * miller_loop(g1[i], g2[i]);
    Fq12 acc = Fq12::one();
    bool started = false;
    for (int i = FINAL_EXP_LIMBS * 64 - 1; i >= 0; i--) {
        if (started) acc = acc * acc;
        if ((FINAL_EXP[i >> 6] >> (i & 63)) & 1) { acc = started ? acc * f : f; started = true; }
    }
    return acc.is_one();
}

}  // namespace host
}  // namespace plk
