#pragma once
#include "ctx.h"
namespace plk {
Fr ntt_omega(uint32_t log_n);
Fr cached_inverse(plk_ctx *ctx, const Fr &g);
int32_t ntt_init_tables(plk_ctx *ctx);
int32_t ntt_coset_table(plk_ctx *ctx, const Fr &g, PowTable *out);
// in-place, natural order in/out; coset may be null
int32_t ntt_dev(plk_ctx *ctx, Fr *data, uint32_t log_n, bool inverse, const Fr *coset, hipStream_t stream);
// n coefficients -> 4n evaluations on coset 7*<omega_4n>
int32_t lde4_dev(plk_ctx *ctx, const Fr *coeffs, uint32_t log_n, Fr *out_4n, hipStream_t stream);
}  // namespace plk
