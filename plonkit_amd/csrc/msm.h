#pragma once
#include "ctx.h"
#include "hostmath.h"
namespace plk {
// enqueue the whole Pippenger pipeline on `stream`; results (window sums) are copied to ctx->pinned
int32_t msm_enqueue(plk_ctx *ctx, const Fr *scalars_dev, uint64_t n, uint64_t base_offset, hipStream_t stream);
// synchronise and fold the window sums on the host
int32_t msm_finish(plk_ctx *ctx, hipStream_t stream, host::HJac *out);
// up to 8 scalar vectors of the same length against the same bases: one pass of every kernel
int32_t msm_enqueue_batch(plk_ctx *ctx, const Fr *const *scalars_dev, uint32_t batch, uint64_t n, uint64_t base_offset, hipStream_t stream);
int32_t msm_finish_batch(plk_ctx *ctx, hipStream_t stream, host::HJac *out);
int32_t ensure_pinned(plk_ctx *ctx, size_t bytes);
}  // namespace plk
