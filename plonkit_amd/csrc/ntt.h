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
// `count` transforms of the same shape, one launch per pass; lane = which ping-pong scratch (0: the context's main
// stream, 1: the prover's background stream — transforms enqueued on two streams at once must not share a scratch)
int32_t ntt_batch_dev(plk_ctx *ctx, Fr *const *data, uint32_t count, uint32_t log_n, bool inverse, const Fr *coset, hipStream_t stream, uint32_t lane);
// the 4n evaluations in coset-major order: out[k * n + r] = f(7 * omega_4n^(4 r + k)) (four n-point coset transforms; prover-internal layout)
int32_t lde4cm_batch_dev(plk_ctx *ctx, const Fr *const *coeffs, uint32_t count, uint32_t log_n, Fr *const *out_4n, hipStream_t stream, uint32_t lane);
// coset-major 4n values -> per coset k the n coefficients u_k (in place, u_k at data + k*n); poly.hip icoset_combine finishes the coset iNTT
int32_t icoset4cm_dev(plk_ctx *ctx, Fr *data_4n, uint32_t log_n, hipStream_t stream, uint32_t lane);
int32_t lde4_batch_dev(plk_ctx *ctx, const Fr *const *coeffs, uint32_t count, uint32_t log_n, Fr *const *out_4n, hipStream_t stream, uint32_t lane);
}  // namespace plk
