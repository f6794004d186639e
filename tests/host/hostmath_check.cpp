// -m "not gpu": the host field arithmetic around the kernels (plonkit_amd/csrc/hostmath.h) — the binary-Euclid inverse against the Fermat
// exponentiation it replaced, and the shortened square-and-multiply, on random and edge values of both fields.  Prints "hostmath ok".
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include "hostmath.h"
using namespace plk::host;

static uint64_t sm64(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }

// the looped six-limb form operator* had until round 5: an independent statement of the same product
template <class T, class PR> static T mul_looped(const T &a, const T &b) {
    typedef unsigned __int128 u128;
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * PR::INV;
        c = ((u128)m * PR::P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * PR::P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    if (t[4] || T::geq_p(t)) T::sub_p(t);
    T r; memcpy(r.l, t, 32); return r;
}

template <class T, class PR> static int run(const char *name) {
    uint64_t seed = 0x706c6b;
    int bad = 0;
    {   // products: random pairs and the extreme stored values (p - 1, 1, 0, all limbs high)
        T ex[5]; ex[0] = T::zero(); ex[1] = T::one();
        { uint64_t c[4] = {PR::P[0] - 1, PR::P[1], PR::P[2], PR::P[3]}; memcpy(ex[2].l, c, 32); }
        { uint64_t c[4] = {1, 0, 0, 0}; memcpy(ex[3].l, c, 32); }
        { uint64_t c[4] = {~0ull, ~0ull, ~0ull, PR::P[3] - 1}; memcpy(ex[4].l, c, 32); }
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) if (!(ex[i] * ex[j] == mul_looped<T, PR>(ex[i], ex[j]))) bad++;
        for (int k = 0; k < 200000; k++) {
            T a, b;
            for (int i = 0; i < 4; i++) { a.l[i] = sm64(seed); b.l[i] = sm64(seed); }
            a.l[3] >>= 3; b.l[3] >>= 3;
            if (T::geq_p(a.l) || T::geq_p(b.l)) continue;
            if (!(a * b == mul_looped<T, PR>(a, b))) bad++;
            if (k < 5) for (int i = 0; i < 5; i++) if (!(a * ex[i] == mul_looped<T, PR>(a, ex[i])) || !(ex[i] * a == mul_looped<T, PR>(ex[i], a))) bad++;
        }
    }
    // inv() is the division-step inverse of round 6 (verified inside, Euclid as its fallback): it must equal the Fermat exponentiation AND the
    // Euclid routine, and its own path must be the one that answered — inv_divsteps never gives up and its raw result already passes the check
    static const T R3 = [] { T r2; memcpy(r2.l, PR::R2, 32); return r2 * r2; }();
    int fallbacks = 0;
    auto check = [&](const T &a) {
        const T i1 = a.inv(), i2 = a.inv_fermat(), i3 = a.inv_euclid();
        if (!(i1 == i2) || !(i1 == i3)) bad++;
        if (!a.is_zero() && !(a * i1 == T::one())) bad++;
        if (!a.is_zero()) {
            T r;
            if (!T::inv_divsteps(a.l, r.l) || !(a * (r * R3) == T::one())) fallbacks++;
        }
    };
    check(T::zero()); check(T::one()); check(-T::one()); check(T::from_u64(2)); check(-T::from_u64(2));
    { T t; uint64_t c[4] = {PR::P[0] - 1, PR::P[1], PR::P[2], PR::P[3]}; memcpy(t.l, c, 32); check(t); }      // stored limbs p - 1
    { T t; uint64_t c[4] = {1, 0, 0, 0}; memcpy(t.l, c, 32); check(t); }                                       // stored limbs 1 (= R^-1)
    for (int k = 0; k < 64; k++) { T t = T::zero(); t.l[k >> 4] = 1ull << ((k & 15) * 4); if (!T::geq_p(t.l)) check(t); }   // sparse limbs
    for (int k = 0; k < 40000; k++) {
        uint64_t c[4] = {sm64(seed), sm64(seed), sm64(seed), sm64(seed) >> 3};
        if (T::geq_p(c)) continue;
        check(T::from_canonical(c));
        if (k < 2000) { T t; memcpy(t.l, c, 32); check(t); }                                                  // the same limbs as STORED (Montgomery) value
    }
    for (uint64_t v = 1; v < 3000; v++) { check(T::from_u64(v)); check(-T::from_u64(v)); T t = T::zero(); t.l[0] = v; check(t); }   // small values, p - small, small stored limbs
    // pow: the short loop against repeated multiplication, and exponents with a high top bit
    for (int k = 0; k < 50; k++) {
        uint64_t c[4] = {sm64(seed), sm64(seed), sm64(seed), sm64(seed) >> 3};
        if (T::geq_p(c)) continue;
        const T a = T::from_canonical(c);
        T acc = T::one();
        for (uint64_t e = 0; e < 40; e++) { if (!(a.pow_u64(e) == acc)) bad++; acc = acc * a; }
        T sq = a; for (int i = 0; i < 20; i++) sq = sq.sqr();
        if (!(a.pow_u64(1ull << 20) == sq)) bad++;
        uint64_t pm1[4] = {PR::P[0] - 1, PR::P[1], PR::P[2], PR::P[3]};
        if (!a.is_zero() && !(a.pow(pm1) == T::one())) bad++;
        uint64_t top[4] = {0, 0, 0, 1ull << 63};
        T t2 = a; for (int i = 0; i < 255; i++) t2 = t2.sqr();
        if (!(a.pow(top) == t2)) bad++;
    }
    T a = T::from_u64(123456789);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 2000; k++) a = a.inv() + T::one();
    auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < 2000; k++) a = a.inv_fermat() + T::one();
    auto t2 = std::chrono::steady_clock::now();
    for (int k = 0; k < 2000; k++) a = a.inv_euclid() + T::one();
    auto t3 = std::chrono::steady_clock::now();
    printf("%s: %d mismatches; %d fallbacks of the division-step inverse; inverse %.2f us (division steps) against %.2f us (binary Euclid) and %.2f us (Fermat)\n", name, bad, fallbacks,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t3 - t2).count() / 2000,
           std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    return bad + fallbacks;
}

int main() {
    int bad = run<HFr, FrP>("Fr") + run<HFq, FqP>("Fq");
    if (bad) { printf("hostmath FAILED\n"); return 1; }
    printf("hostmath ok\n");
    return 0;
}
