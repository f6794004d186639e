// Index plan of the wave-owned NTT tile passes (ntt.hip: ntt_pass_w) — who holds what, and where it goes in LDS.
//
// A workgroup (8 waves x 64 lanes) owns a tile of R = 2^LR rows x C = 2^(11-LR) columns (2048 elements) and runs the
// LR butterfly stages of a DIT transform over the rows (input rows in bit-reversed order) as NR = ceil(LR/2) ROUNDS of
// one or two stages; in every round a lane holds the four rows  i0 + {0, h, 2h, 3h}  (h = 2^s, s the round's first stage)
// of one column in registers.  What changes against the barrier-per-round kernel (ntt_pass_cols / ntt_pass_rows):
//   * the first round is fed from HBM and the last one stores to HBM — a pass has NR - 1 LDS exchanges, not NR + 1;
//   * the tile is split among the waves so that an exchange stays inside ONE wave wherever the stages allow it:
//       phase A (stages < A):  wave w owns the rows whose index bits [A, A+3) equal w — closed under the stages below A,
//       phase B (stages >= A): wave w owns the rows whose index bits [0, 3)   equal w — closed under the stages >= 3.
//     Inside a phase an exchange needs no workgroup barrier (the LDS executes one wave's instructions in order); the
//     only workgroup-wide synchronisation of a pass is the hand-over from phase A to phase B.
//   * LDS is addressed the way the READER wants it: the layout of round r is  slot = wave*4*KS + k*KS + lane  (k = which of
//     the lane's four rows), so every read of a round is 64 consecutive 16-byte words (conflict-free, one address register
//     and immediate offsets) and the writer scatters.  KS = 64 + a small pad chosen per round so that the scattered
//     16-byte stores of eight neighbouring lanes fall into eight different bank groups (tests/host/ntt_plan_check.cpp
//     enumerates every round of every shape: ownership, coverage, exchange correctness, bank conflicts).
// Everything here is constexpr / host-device so that the host test runs the same code the kernel runs.
#pragma once
#include <stdint.h>
#ifndef PLK_HD
#define PLK_HD inline
#endif

namespace plk {

constexpr int NTT_LOG_TILE = 11;
constexpr int NTT_W_KS_MAX = 68, NTT_W_WP_MAX = 8;             // largest padded k-stride / wave pad of any round
constexpr int NTT_W_SLOTS = 8 * (4 * NTT_W_KS_MAX + NTT_W_WP_MAX);   // LDS slots of a tile (36 bytes each): 80 640 B, two tiles per CU
// Round 6: the same plan for a tile of 4096 elements owned by SIXTEEN waves (one workgroup per CU, 161 280 of the CU's 163 840 bytes of LDS): 11-bit digits,
// so that a 2^21 / 2^22-point transform is two passes instead of three.  LT = log2 of the tile; a wave still holds 256 elements (four per lane).
constexpr int NTT_LOG_TILE_BIG = 12;
constexpr int NTT_W_SLOTS_BIG = 16 * (4 * NTT_W_KS_MAX + NTT_W_WP_MAX);

PLK_HD uint32_t plan_brev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

template <int LR, int LT = NTT_LOG_TILE>
struct TilePlan {
    static_assert((LT == NTT_LOG_TILE && LR >= 7 && LR <= 10) || (LT == NTT_LOG_TILE_BIG && LR >= 10 && LR <= 11), "wave-owned passes: 7..10 row bits of a 2048-element tile, 10..11 of a 4096-element one");
    static constexpr int LC = LT - LR;                         // log2 columns
    static constexpr int LWR = LT - 8;                         // 8 (16) waves = 8 (16) row groups
    static constexpr int NW = 1 << LWR;
    static constexpr int SLOTS = NW * (4 * NTT_W_KS_MAX + NTT_W_WP_MAX);
    static constexpr int LRA = LR - LWR;                       // row bits inside a wave (LRA + LC = 8)
    static constexpr int NR = (LR + 1) / 2;                    // rounds
    // first stage / number of stages of round r: an odd LR starts with ONE twiddle-free stage
    static constexpr int rs(int r) { return (LR & 1) ? (r == 0 ? 0 : 2 * r - 1) : 2 * r; }
    static constexpr int rn(int r) { return ((LR & 1) && r == 0) ? 1 : 2; }
    static constexpr int split() { int a = 0; for (int r = 1; r < NR; r++) if (rs(r) <= LR - LWR) a = rs(r); return a; }
    static constexpr int A = split();                          // stages < A: phase A, the rest: phase B
    static constexpr bool pb(int r) { return rs(r) >= A; }
    static constexpr int ls(int r) { return pb(r) ? rs(r) - LWR : rs(r); }     // the stage in wave-local row coordinates
    // strides of the layout round r READS (r >= 1): k blocks of KS(r) = 64 + pad slots inside wave regions of WS = 272 + pad
    // slots.  WS is ONE value per shape: inside a phase wave w may already be writing the layout of round r+1 while wave w'
    // still reads the layout of round r, so the regions of different waves must be disjoint across rounds.  The writer of
    // a layout is the round before it: eight neighbouring lanes of it land in different k blocks and / or different wave
    // regions (hand-over), and the pads spread those over the eight 16-byte bank groups.  `ntt_plan_check search` prints the
    // candidates; the check fails if a store instruction needs more LDS-array cycles than its transfer hides.
    static constexpr int pad_tab(int r);
    static constexpr int KS(int r) { return 64 + pad_tab(r); }
    static constexpr int WS = 272 + pad_tab(0);

    // the (row, col) a lane's register k holds in round r
    template <int r>
    static PLK_HD void locate(uint32_t wave, uint32_t lane, uint32_t k, uint32_t &row, uint32_t &col) {
        constexpr int s = ls(r), h = 1 << s;
        col = lane & ((1u << LC) - 1);
        const uint32_t j = lane >> LC;                                              // LRA - 2 bits
        const uint32_t ix = ((j >> s) << (s + 2)) | (j & (h - 1)) | (k << s);      // wave-local row
        if (pb(r)) row = (ix << LWR) | wave;
        else row = ((ix >> A) << (A + LWR)) | (wave << A) | (ix & ((1u << A) - 1));
    }
    // where the reader of round r expects (row, col)
    template <int r>
    static PLK_HD uint32_t slot(uint32_t row, uint32_t col, uint32_t ks = KS(r), uint32_t ws = WS) {
        constexpr int s = ls(r), h = 1 << s;
        uint32_t wave, ix;
        if (pb(r)) { wave = row & (uint32_t)(NW - 1); ix = row >> LWR; }
        else { wave = (row >> A) & (uint32_t)(NW - 1); ix = ((row >> (A + LWR)) << A) | (row & ((1u << A) - 1)); }
        const uint32_t k = (ix >> s) & 3u, j = ((ix >> (s + 2)) << s) | (ix & (h - 1));
        return wave * ws + k * ks + ((j << LC) | col);
    }
    // what a lane reads in round r: slot(locate(r, wave, lane, k)) without the detour
    template <int r>
    static PLK_HD uint32_t own_slot(uint32_t wave, uint32_t lane, uint32_t k, uint32_t ks = KS(r), uint32_t ws = WS) { return wave * ws + k * ks + lane; }
};

// entry 0: wave stride - 272; entry r >= 1: k pad of the layout round r reads
template <int LR, int LT> constexpr int TilePlan<LR, LT>::pad_tab(int r) {
    constexpr int tab[4][5] = {
        /* LR = 7  */ {0, 0, 0, 0, 0},
        /* LR = 8  */ {8, 4, 0, 0, 0},
        /* LR = 9  */ {4, 2, 4, 0, 0},
        /* LR = 10 */ {2, 2, 4, 0, 0},
    };
    constexpr int big[2][6] = {                                // 4096-element tile (`ntt_plan_check search` prints the candidates)
        /* LR = 10 */ {4, 4, 0, 0, 0, 0},
        /* LR = 11 */ {2, 1, 4, 0, 0, 0},
    };
    return LT == NTT_LOG_TILE ? tab[LR - 7][r < 5 ? r : 4] : big[LR - 10][r];
}

}  // namespace plk
