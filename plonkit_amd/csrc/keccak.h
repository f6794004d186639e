#pragma once
#include <stdint.h>
#include <stddef.h>
#include "hostmath.h"
namespace plk {
void keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
// RollingKeccakTranscript (contrib/template.sol:267-307)
struct RollingKeccak {
    uint8_t s0[32] = {0}, s1[32] = {0};
    uint32_t counter = 0;
    void absorb_word(const uint8_t w[32]);
    void absorb_fr(const host::HFr &v);
    void absorb_g1(const host::HAffine &p);
    host::HFr challenge();
};
void g1_to_bytes(const host::HAffine &p, uint8_t out[64]);
bool g1_from_bytes(const uint8_t in[64], host::HAffine *out);
}  // namespace plk
