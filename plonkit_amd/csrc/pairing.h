// Host-side BN254 optimal-ate pairing, used only by the verifier (`plk_verify`, CLI `verify`).
// Replaces pairing_ce::bn256's Engine::final_exponentiation(miller_loop(..)) as reached from
// bellman_ce::plonk::better_cs::verifier::verify (reference call site: src/plonk.rs:189-210).
// The verifier is CPU code in the reference as well; nothing here touches the GPU.
//
// Representation (chosen for auditability over speed — one verification is ~5000 Fq12 products, 20 ms):
//   Fq2  = Fq[i]/(i^2 + 1)
//   Fq12 = Fq[w]/(w^12 - 18 w^6 + 82)   — a flat degree-12 ring; i = w^6 - 9, so xi = 9 + i = w^6
//   G2 lives on the sextic twist E': y^2 = x^3 + 3/xi over Fq2; (x, y) -> (x w^2, y w^3) lands on E(Fq12).
// The value of one pairing is not compared with anything outside; the verifier only needs the product check
// e(A, Q0) * e(B, Q1) == 1, which any bilinear non-degenerate pairing decides identically.
#pragma once
#include "hostmath.h"

namespace plk {
namespace host {

struct Fq2 {
    HFq c0, c1;
    static Fq2 zero() { return {HFq::zero(), HFq::zero()}; }
    static Fq2 one() { return {HFq::one(), HFq::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fq2 &o) const { return c0 == o.c0 && c1 == o.c1; }
    Fq2 operator+(const Fq2 &o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fq2 operator-(const Fq2 &o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fq2 operator-() const { return {-c0, -c1}; }
    Fq2 operator*(const Fq2 &o) const { return {c0 * o.c0 - c1 * o.c1, c0 * o.c1 + c1 * o.c0}; }
    Fq2 scale(const HFq &k) const { return {c0 * k, c1 * k}; }
    Fq2 sqr() const { return *this * *this; }
    Fq2 conj() const { return {c0, -c1}; }
    Fq2 inv() const { HFq n = (c0.sqr() + c1.sqr()).inv(); return {c0 * n, -(c1 * n)}; }
};

struct G2Affine {
    Fq2 x, y;
    bool inf;
};

struct Fq12 {
    HFq c[12];
    static Fq12 zero();
    static Fq12 one();
    bool is_one() const;
    Fq12 operator*(const Fq12 &o) const;
};

// x.c1 || x.c0 || y.c1 || y.c0, 32-byte big-endian canonical each; checks the twist equation
bool g2_from_bytes(const uint8_t in[128], G2Affine *out);

// product of Miller loops over the pairs, then one final exponentiation; true when the result is 1
bool pairing_product_is_one(const HAffine *g1, const G2Affine *g2, int pairs);

}  // namespace host
}  // namespace plk
