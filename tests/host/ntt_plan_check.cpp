// Host-side enumeration of the wave-owned NTT tile plan (plonkit_amd/csrc/ntt_plan.h — the same constexpr code the kernel
// ntt_pass_w runs): for every shape LR = 7..10 and every round
//   * the 8 x 64 x 4 (wave, lane, k) registers of a round cover the 2048 (row, col) of the tile exactly once, and a lane's
//     four rows are i0 + {0, h, 2h, 3h} with the stage's pairing;
//   * the exchange writer(r-1) -> reader(r) through slot<r>() hands every lane exactly the rows locate<r>() says it holds;
//   * inside a phase a wave only writes its own LDS region (that is what makes a wave-local synchronisation enough);
//   * LDS bank conflicts of the scattered stores with the padded k-strides of the plan (and the best pad found by search).
// Built with g++ and run by tests/test_ntt_plan_host.py; `ntt_plan_check search` prints the pad table.  Needs no GPU.
#include "ntt_plan.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <utility>
using namespace plk;

static int bad = 0;
#define CHECK(c, ...) do { if (!(c)) { if (bad < 20) { printf("FAIL: "); printf(__VA_ARGS__); printf("\n"); } bad++; } } while (0)

// array cycles of one ds_write_b128 wave instruction: 8 groups of 8 consecutive lanes, bank group = slot mod 8
static int b128_write_cycles(const uint32_t slot[64]) {
    int cyc = 0;
    for (int g = 0; g < 8; g++) { int cnt[8] = {0}; int mx = 0; for (int l = 0; l < 8; l++) { int b = slot[8 * g + l] & 7; if (++cnt[b] > mx) mx = cnt[b]; } cyc += mx; }
    return cyc;
}
// ds_write_b32: 2 groups of 32 lanes, bank = slot mod 32
static int b32_write_cycles(const uint32_t slot[64]) {
    int cyc = 0;
    for (int g = 0; g < 2; g++) { int cnt[32] = {0}; int mx = 0; for (int l = 0; l < 32; l++) { int b = slot[32 * g + l] & 31; if (++cnt[b] > mx) mx = cnt[b]; } cyc += mx; }
    return cyc;
}

template <int LR, int LT, int r>
static void check_round(bool search) {
    using P = TilePlan<LR, LT>;
    constexpr uint32_t NW = P::NW;
    constexpr int R = 1 << LR, C = 1 << P::LC;
    // coverage + pairing of round r
    std::vector<int> seen(R * C, 0);
    for (uint32_t w = 0; w < NW; w++) for (uint32_t l = 0; l < 64; l++) {
        uint32_t row[4], col[4];
        for (uint32_t k = 0; k < 4; k++) { P::template locate<r>(w, l, k, row[k], col[k]); CHECK(row[k] < (uint32_t)R && col[k] < (uint32_t)C, "LR %d round %d out of range", LR, r); seen[row[k] * C + col[k]]++; }
        const uint32_t h = 1u << P::rs(r);
        CHECK(col[0] == col[1] && col[1] == col[2] && col[2] == col[3], "LR %d round %d columns differ", LR, r);
        CHECK(row[1] == row[0] + h && row[2] == row[0] + 2 * h && row[3] == row[0] + 3 * h, "LR %d round %d rows not i0 + k h", LR, r);
        CHECK((row[0] & (3 * h)) == 0, "LR %d round %d i0 has stage bits set", LR, r);
    }
    for (int i = 0; i < R * C; i++) CHECK(seen[i] == 1, "LR %d round %d (row %d col %d) held %d times", LR, r, i / C, i % C, seen[i]);
    if constexpr (r > 0) {
        int best_kp = -1, best_wp = -1, best_cyc = 1 << 30;
        for (int pi = -1; pi < (search ? 5 * 9 : 0); pi++) {
            const int kp = pi < 0 ? P::KS(r) - 64 : pi % 5, wp = pi < 0 ? P::WS - 272 : pi / 5;
            const uint32_t ks = 64 + kp, ws = 272 + wp;                         // one wave stride for every round of a shape
            std::vector<uint32_t> lds(P::SLOTS, 0xffffffffu);
            int cyc128 = 0, cyc32 = 0, n_inst = 0;
            for (uint32_t w = 0; w < NW; w++) for (uint32_t k = 0; k < 4; k++) {
                uint32_t sl[64];
                for (uint32_t l = 0; l < 64; l++) {
                    uint32_t row, col;
                    P::template locate<r - 1>(w, l, k, row, col);
                    sl[l] = P::template slot<r>(row, col, ks, ws);
                    CHECK(sl[l] < (uint32_t)P::SLOTS, "LR %d round %d slot %u out of range", LR, r, sl[l]);
                    if (P::pb(r - 1) == P::pb(r)) CHECK(sl[l] / ws == w, "LR %d round %d: wave %u writes the region of wave %u inside a phase", LR, r, w, sl[l] / ws);
                    CHECK(lds[sl[l]] == 0xffffffffu, "LR %d round %d slot %u written twice", LR, r, sl[l]);
                    lds[sl[l]] = row << 8 | col;
                }
                cyc128 += b128_write_cycles(sl); cyc32 += b32_write_cycles(sl); n_inst++;
            }
            for (uint32_t w = 0; w < NW; w++) for (uint32_t l = 0; l < 64; l++) for (uint32_t k = 0; k < 4; k++) {
                uint32_t row, col;
                P::template locate<r>(w, l, k, row, col);
                CHECK(lds[P::template own_slot<r>(w, l, k, ks, ws)] == (row << 8 | col), "LR %d round %d: wave %u lane %u k %u reads the wrong element", LR, r, w, l, k);
            }
            if (pi < 0) printf("%sLR %2d round %d (stages %d..%d, %s): k pad %d, wave stride 272 + %d:  ds_write_b128 array cycles %.2f (8 = conflict-free, <= 13 hidden by the transfer)  ds_write_b32 %.2f (2 = conflict-free)\n",
                               LT == NTT_LOG_TILE ? "" : "tile 4096, ", LR, r, P::rs(r), P::rs(r) + P::rn(r) - 1, P::pb(r - 1) != P::pb(r) ? "A->B hand-over" : (P::pb(r) ? "phase B" : "phase A"),
                               kp, wp, (double)cyc128 / n_inst, (double)cyc32 / n_inst);
            if (pi < 0) CHECK(ks * 4 <= ws && NW * ws <= (uint32_t)P::SLOTS, "LR %d round %d: strides %u / %u do not fit", LR, r, ks, ws);
            if (pi < 0 && !search) CHECK(cyc128 <= 13 * n_inst && cyc32 <= 4 * n_inst, "LR %d round %d: stores conflict beyond the transfer time (%.2f / %.2f array cycles)", LR, r, (double)cyc128 / n_inst, (double)cyc32 / n_inst);
            if (pi >= 0) { if (kp == 0) printf("      ws %d:", ws); if (cyc128 == 8 * n_inst) printf(" kp %d (b32 %.0f)", kp, (double)cyc32 / n_inst); if (kp == 4) printf("\n"); }
        }
        (void)best_kp; (void)best_wp; (void)best_cyc;
    }
}
template <int LR, int LT, int... r> static void check_rounds(bool search, std::integer_sequence<int, r...>) { (check_round<LR, LT, r>(search), ...); }
template <int LR, int LT = NTT_LOG_TILE> static void check_shape(bool search) {
    using P = TilePlan<LR, LT>;
    printf("%sLR %d: C %d, rounds %d, phase split A = %d\n", LT == NTT_LOG_TILE ? "" : "tile 4096, ", LR, 1 << P::LC, P::NR, P::A);
    CHECK(P::A >= P::LWR && P::A <= LR - P::LWR, "LR %d: split %d outside [3, LR - 3]", LR, P::A);
    CHECK(P::rs(P::NR - 1) + P::rn(P::NR - 1) == LR, "LR %d: rounds do not end at LR", LR);
    check_rounds<LR, LT>(search, std::make_integer_sequence<int, P::NR>{});
}
int main(int argc, char **argv) {
    const bool search = argc > 1 && !strcmp(argv[1], "search");
    check_shape<7>(search); check_shape<8>(search); check_shape<9>(search); check_shape<10>(search);
    check_shape<10, NTT_LOG_TILE_BIG>(search); check_shape<11, NTT_LOG_TILE_BIG>(search);
    if (bad) { printf("%d failures\n", bad); return 1; }
    printf("ntt plan ok\n");
    return 0;
}
