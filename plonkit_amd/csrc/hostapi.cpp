// CPU-only entry points of the C ABI: point/scalar (de)serialisation in the reference's file
// encoding (big-endian canonical, infinity = 0x40 00..), Ethereum keccak-256 and
// RollingKeccakTranscript (src/plonk.rs:10,140,152; byte recipe contrib/template.sol:267-307).
#include "../../include/plonkit_amd.h"
#include "hostmath.h"
#include "keccak.h"
#include <string>

namespace plk { void set_error(const std::string &msg); }
using namespace plk;
using namespace plk::host;

#define PLK_API extern "C" __attribute__((visibility("default")))

namespace plk {

static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static inline uint64_t rotl(uint64_t x, unsigned s) { return s ? (x << s) | (x >> (64 - s)) : x; }

static void keccak_permute(uint64_t a[25]) {
    // rho offsets generated along the pi cycle starting at lane (1,0)
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            uint64_t d = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
            for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
        }
        int x = 1, y = 0;
        uint64_t cur = a[1];
        for (int t = 0; t < 24; t++) {
            int nx = y, ny = (2 * x + 3 * y) % 5;
            uint64_t tmp = a[nx + 5 * ny];
            a[nx + 5 * ny] = rotl(cur, ((t + 1) * (t + 2) / 2) % 64);
            cur = tmp; x = nx; y = ny;
        }
        for (int yy = 0; yy < 25; yy += 5) {
            uint64_t r0 = a[yy], r1 = a[yy + 1], r2 = a[yy + 2], r3 = a[yy + 3], r4 = a[yy + 4];
            a[yy] = r0 ^ (~r1 & r2); a[yy + 1] = r1 ^ (~r2 & r3); a[yy + 2] = r2 ^ (~r3 & r4);
            a[yy + 3] = r3 ^ (~r4 & r0); a[yy + 4] = r4 ^ (~r0 & r1);
        }
        a[0] ^= KRC[rnd];
    }
}

void keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
    uint64_t st[25] = {0};
    const size_t rate = 136;
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, in + 8 * i, 8); st[i] ^= w; }
        keccak_permute(st);
        in += rate; len -= rate;
    }
    uint8_t last[136] = {0};
    memcpy(last, in, len);
    last[len] ^= 0x01;            // Ethereum keccak padding, not SHA3's 0x06
    last[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) { uint64_t w; memcpy(&w, last + 8 * i, 8); st[i] ^= w; }
    keccak_permute(st);
    memcpy(out, st, 32);
}

void RollingKeccak::absorb_word(const uint8_t w[32]) {
    uint8_t buf[4 + 32 + 32 + 32];
    memcpy(buf + 4, s0, 32); memcpy(buf + 36, s1, 32); memcpy(buf + 68, w, 32);
    uint8_t n0[32], n1[32];
    buf[0] = buf[1] = buf[2] = 0; buf[3] = 0;
    keccak256(buf, sizeof buf, n0);
    buf[3] = 1;
    keccak256(buf, sizeof buf, n1);
    memcpy(s0, n0, 32); memcpy(s1, n1, 32);
}

void RollingKeccak::absorb_fr(const HFr &v) { uint8_t w[32]; v.to_be_bytes(w); absorb_word(w); }

void RollingKeccak::absorb_g1(const HAffine &p) {
    uint8_t w[32];
    p.x.to_be_bytes(w); absorb_word(w);       // infinity is (0, 0) in memory and is absorbed as (0, 0)
    p.y.to_be_bytes(w); absorb_word(w);
}

HFr RollingKeccak::challenge() {
    uint8_t buf[4 + 32 + 32 + 4], q[32];
    buf[0] = buf[1] = buf[2] = 0; buf[3] = 2;
    memcpy(buf + 4, s0, 32); memcpy(buf + 36, s1, 32);
    buf[68] = (uint8_t)(counter >> 24); buf[69] = (uint8_t)(counter >> 16); buf[70] = (uint8_t)(counter >> 8); buf[71] = (uint8_t)counter;
    counter++;
    keccak256(buf, sizeof buf, q);
    q[0] &= 0x1f;                             // keep the low 253 bits
    HFr r;
    HFr::from_be_bytes(q, &r);                // < 2^253 < r, always valid
    return r;
}

void g1_to_bytes(const HAffine &p, uint8_t out[64]) {
    if (p.is_inf()) { memset(out, 0, 64); out[0] = 0x40; return; }
    p.x.to_be_bytes(out); p.y.to_be_bytes(out + 32);
}

bool g1_from_bytes(const uint8_t in[64], HAffine *out) {
    if (in[0] & 0x40) { out->x = HFq::zero(); out->y = HFq::zero(); return true; }
    return HFq::from_be_bytes(in, &out->x) && HFq::from_be_bytes(in + 32, &out->y);
}

}  // namespace plk

PLK_API void plk_keccak256(const uint8_t *in, uint64_t len, uint8_t out[32]) { keccak256(in, len, out); }

PLK_API void plk_transcript_init(plk_transcript *t) { memset(t, 0, sizeof *t); }

static RollingKeccak load_tr(const plk_transcript *t) { RollingKeccak r; memcpy(r.s0, t->state0, 32); memcpy(r.s1, t->state1, 32); r.counter = t->counter; return r; }
static void store_tr(plk_transcript *t, const RollingKeccak &r) { memcpy(t->state0, r.s0, 32); memcpy(t->state1, r.s1, 32); t->counter = r.counter; }

PLK_API void plk_transcript_absorb_fr(plk_transcript *t, const plk_fr *v) {
    RollingKeccak r = load_tr(t); HFr f; memcpy(f.l, v->l, 32); r.absorb_fr(f); store_tr(t, r); }
PLK_API void plk_transcript_absorb_g1(plk_transcript *t, const plk_g1_affine *p) {
    RollingKeccak r = load_tr(t); HAffine a; memcpy(a.x.l, p->x, 32); memcpy(a.y.l, p->y, 32); r.absorb_g1(a); store_tr(t, r); }
PLK_API void plk_transcript_challenge(plk_transcript *t, plk_fr *out) {
    RollingKeccak r = load_tr(t); HFr c = r.challenge(); memcpy(out->l, c.l, 32); store_tr(t, r); }

PLK_API int32_t plk_g1_on_curve(const plk_g1_affine *p) {
    if (!p) return 0;
    HAffine a; memcpy(a.x.l, p->x, 32); memcpy(a.y.l, p->y, 32);
    return on_curve(a) ? 1 : 0;
}
PLK_API int32_t plk_g1_to_bytes(const plk_g1_affine *p, uint8_t out[64]) {
    if (!p || !out) { set_error("plk_g1_to_bytes: null"); return PLK_ERR_ARG; }
    HAffine a; memcpy(a.x.l, p->x, 32); memcpy(a.y.l, p->y, 32);
    g1_to_bytes(a, out); return PLK_OK;
}
PLK_API int32_t plk_g1_from_bytes(const uint8_t in[64], plk_g1_affine *out) {
    if (!in || !out) { set_error("plk_g1_from_bytes: null"); return PLK_ERR_ARG; }
    HAffine a;
    if (!g1_from_bytes(in, &a)) { set_error("G1 coordinate not in field"); return PLK_ERR_FORMAT; }
    if (!on_curve(a)) { set_error("G1 point not on curve"); return PLK_ERR_FORMAT; }
    memcpy(out->x, a.x.l, 32); memcpy(out->y, a.y.l, 32); return PLK_OK;
}
PLK_API int32_t plk_fr_to_bytes(const plk_fr *a, uint8_t out[32]) {
    if (!a || !out) { set_error("plk_fr_to_bytes: null"); return PLK_ERR_ARG; }
    HFr f; memcpy(f.l, a->l, 32); f.to_be_bytes(out); return PLK_OK;
}
PLK_API int32_t plk_fr_from_bytes(const uint8_t in[32], plk_fr *out) {
    if (!in || !out) { set_error("plk_fr_from_bytes: null"); return PLK_ERR_ARG; }
    HFr f;
    if (!HFr::from_be_bytes(in, &f)) { set_error("scalar not in field"); return PLK_ERR_FORMAT; }
    memcpy(out->l, f.l, 32); return PLK_OK;
}
