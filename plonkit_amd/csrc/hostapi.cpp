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
    if (len) memcpy(last, in, len);                     // (in may be null for an empty message)
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

// pairing_ce's G1Uncompressed::into_affine_unchecked: the infinity flag must come with an all-zero remainder, and the
// coordinates (0, 0) without the flag are an ordinary (off-curve) point, not infinity — both are refused there
bool g1_from_bytes(const uint8_t in[64], HAffine *out) {
    if (in[0] & 0x40) {
        if (in[0] != 0x40) return false;
        for (int i = 1; i < 64; i++) if (in[i]) return false;
        out->x = HFq::zero(); out->y = HFq::zero();
        return true;
    }
    if (in[0] & 0x80) return false;                       // compression flag on an uncompressed encoding
    if (!HFq::from_be_bytes(in, &out->x) || !HFq::from_be_bytes(in + 32, &out->y)) return false;
    return !(out->x.is_zero() && out->y.is_zero());       // (0, 0) unflagged: not on the curve
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

// ---------------------------------------------------------------- SRS key files (Crs::read / write)
// layout: u64 n_g1 (BE) | n_g1 x G1 (64 B, BE canonical, 0x40.. = infinity) | u64 2 | 2 x G2 (128 B)
#include <thread>
#include <vector>
#include <atomic>

// G2 section of Crs::crs_42 (src/plonk.rs:41,47): {G2 generator, 42 * G2}, x.c1 | x.c0 | y.c1 | y.c0 big-endian —
// BN254 constants, identical to the tail of the reference's keys/setup/setup_2^10.key
static const uint8_t CRS42_G2[256] = {
    0x19, 0x8e, 0x93, 0x93, 0x92, 0x0d, 0x48, 0x3a, 0x72, 0x60, 0xbf, 0xb7, 0x31, 0xfb, 0x5d, 0x25,
    0xf1, 0xaa, 0x49, 0x33, 0x35, 0xa9, 0xe7, 0x12, 0x97, 0xe4, 0x85, 0xb7, 0xae, 0xf3, 0x12, 0xc2,
    0x18, 0x00, 0xde, 0xef, 0x12, 0x1f, 0x1e, 0x76, 0x42, 0x6a, 0x00, 0x66, 0x5e, 0x5c, 0x44, 0x79,
    0x67, 0x43, 0x22, 0xd4, 0xf7, 0x5e, 0xda, 0xdd, 0x46, 0xde, 0xbd, 0x5c, 0xd9, 0x92, 0xf6, 0xed,
    0x09, 0x06, 0x89, 0xd0, 0x58, 0x5f, 0xf0, 0x75, 0xec, 0x9e, 0x99, 0xad, 0x69, 0x0c, 0x33, 0x95,
    0xbc, 0x4b, 0x31, 0x33, 0x70, 0xb3, 0x8e, 0xf3, 0x55, 0xac, 0xda, 0xdc, 0xd1, 0x22, 0x97, 0x5b,
    0x12, 0xc8, 0x5e, 0xa5, 0xdb, 0x8c, 0x6d, 0xeb, 0x4a, 0xab, 0x71, 0x80, 0x8d, 0xcb, 0x40, 0x8f,
    0xe3, 0xd1, 0xe7, 0x69, 0x0c, 0x43, 0xd3, 0x7b, 0x4c, 0xe6, 0xcc, 0x01, 0x66, 0xfa, 0x7d, 0xaa,
    0x12, 0x74, 0x09, 0x34, 0xba, 0x96, 0x15, 0xb7, 0x7b, 0x6a, 0x49, 0xb0, 0x6f, 0xcc, 0xe8, 0x3c,
    0xe9, 0x0d, 0x67, 0xb1, 0xd0, 0xe2, 0xa5, 0x30, 0x06, 0x9e, 0x3a, 0x73, 0x06, 0x56, 0x9a, 0x91,
    0x11, 0x6d, 0xa8, 0xc8, 0x9a, 0x0d, 0x09, 0x0f, 0x3d, 0x86, 0x44, 0xad, 0xa3, 0x3a, 0x5f, 0x1c,
    0x80, 0x13, 0xba, 0x72, 0x04, 0xae, 0xca, 0x62, 0xd6, 0x6d, 0x93, 0x1b, 0x99, 0xaf, 0xe6, 0xe7,
    0x25, 0x22, 0x2d, 0x98, 0x16, 0xe5, 0xf8, 0x6b, 0x4a, 0x7d, 0xed, 0xd0, 0x0d, 0x04, 0xac, 0xc5,
    0xc9, 0x79, 0xc1, 0x8b, 0xd2, 0x2b, 0x83, 0x4e, 0xa8, 0xc6, 0xd0, 0x7c, 0x0b, 0xa4, 0x41, 0xdb,
    0x07, 0x64, 0x41, 0x04, 0x2e, 0x77, 0xb6, 0x30, 0x96, 0x44, 0xb5, 0x62, 0x51, 0xf0, 0x59, 0xcf,
    0x14, 0xbe, 0xfc, 0x72, 0xac, 0x8a, 0x61, 0x57, 0xd3, 0x09, 0x24, 0xe5, 0x8d, 0xc4, 0xc1, 0x72};

PLK_API void plk_crs42_g2_bytes(uint8_t out[256]) { memcpy(out, CRS42_G2, 256); }

PLK_API int32_t plk_key_parse(const uint8_t *data, uint64_t len, plk_g1_affine *points, uint64_t cap, uint64_t *n_out, uint8_t g2_out[256]) {
    if (!data || !n_out) { set_error("plk_key_parse: bad argument"); return PLK_ERR_ARG; }
    if (len < 8) { set_error("read key err: truncated"); return PLK_ERR_FORMAT; }
    uint64_t n = 0;
    for (int i = 0; i < 8; i++) n = (n << 8) | data[i];
    *n_out = n;
    if (n > (1ull << 28) || len < 8 + 64 * n + 8 + 256) { set_error("read key err: truncated"); return PLK_ERR_FORMAT; }
    uint64_t n2 = 0;
    for (int i = 0; i < 8; i++) n2 = (n2 << 8) | data[8 + 64 * n + i];
    if (n2 != 2) { set_error("read key err: expected two G2 points"); return PLK_ERR_FORMAT; }
    if (g2_out) memcpy(g2_out, data + 8 + 64 * n + 8, 256);
    if (!points) return PLK_OK;                       // size query
    if (cap < n) { set_error("plk_key_parse: point buffer too small"); return PLK_ERR_ARG; }
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    std::atomic<int> bad(0);
    std::vector<std::thread> th;
    uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        uint64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        th.emplace_back([&, lo, hi]() {
            for (uint64_t i = lo; i < hi; i++) {
                HAffine a;
                if (!g1_from_bytes(data + 8 + 64 * i, &a) || !on_curve(a)) { bad = 1; return; }   // Crs::read checks every point
                memcpy(points[i].x, a.x.l, 32); memcpy(points[i].y, a.y.l, 32);
            }
        });
    }
    for (auto &x : th) x.join();
    if (bad) { set_error("read key err: point not on curve"); return PLK_ERR_FORMAT; }
    return PLK_OK;
}

PLK_API int32_t plk_key_serialize(const plk_g1_affine *points, uint64_t n, const uint8_t g2[256], uint8_t *out, uint64_t cap, uint64_t *len) {
    if (!points || !g2 || !len) { set_error("plk_key_serialize: bad argument"); return PLK_ERR_ARG; }
    *len = 8 + 64 * n + 8 + 256;
    if (!out) return PLK_OK;
    if (cap < *len) { set_error("plk_key_serialize: buffer too small"); return PLK_ERR_ARG; }
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(n >> (8 * (7 - i)));
    unsigned nt = std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    std::vector<std::thread> th;
    uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        uint64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        th.emplace_back([=]() {
            for (uint64_t i = lo; i < hi; i++) { HAffine a; memcpy(a.x.l, points[i].x, 32); memcpy(a.y.l, points[i].y, 32); g1_to_bytes(a, out + 8 + 64 * i); }
        });
    }
    for (auto &x : th) x.join();
    uint8_t *p = out + 8 + 64 * n;
    for (int i = 0; i < 7; i++) p[i] = 0;
    p[7] = 2;
    memcpy(p + 8, g2, 256);
    return PLK_OK;
}
