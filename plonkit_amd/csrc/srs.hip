// SRS generation on the GPU: Crs::<Bn256, CrsForMonomialForm>::crs_42(size, &Worker)
// (src/plonk.rs:30-48 `gen_key_monomial_form`; `plonkit setup`, src/bin/main.rs:334-343).
// g1[i] = tau^i * G with the reference's insecure tau = 42 (SURVEY.md A.1 [derived]) — N fixed-base
// scalar multiplications.  Each lane owns a run of consecutive powers: one double-and-add to reach
// tau^(start) * G, then "multiply by tau" steps, each converted to affine with one Fermat inversion.
#include "ctx.h"
#include "ec_dev.h"
#include <cstring>

namespace plk {

constexpr uint32_t SRS_RUN = 16;

__device__ __forceinline__ G1Affine xyzz_to_affine_dev(const G1Xyzz &p) {
    G1Affine a;
    if (is_inf(p)) { a.x = Fq::zero(); a.y = Fq::zero(); return a; }
    Fq i = inv(mul(p.zz, p.zzz));           // 1/(zz*zzz): 1/zz = i*zzz, 1/zzz = i*zz
    a.x = mul(p.x, mul(i, p.zzz));
    a.y = mul(p.y, mul(i, p.zz));
    return a;
}

__global__ void __launch_bounds__(256) srs_powers_kernel(G1Affine *out, uint64_t start, uint64_t n, Fr tau, uint32_t tau_small) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i0 = t * SRS_RUN;
    if (i0 >= n) return;
    Fr k = to_canonical(pow_u64(tau, start + i0));
    G1Affine g; g.x = from_u64<FqParams>(1); g.y = from_u64<FqParams>(2);
    G1Xyzz p = xyzz_identity();
    for (int bit = 253; bit >= 0; bit--) {
        p = xyzz_double(p);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzz_add_mixed(p, g, false);
    }
    for (uint32_t j = 0; j < SRS_RUN && i0 + j < n; j++) {
        G1Affine a = xyzz_to_affine_dev(p);
        store_fp(&out[i0 + j].x, a.x);
        store_fp(&out[i0 + j].y, a.y);
        p = xyzz_mul_small(p, tau_small);
    }
}

// general tau (any field element): one full double-and-add per point, no run sharing — 16x the work of the small-tau kernel,
// still a fraction of a second at 2^20 points
__global__ void __launch_bounds__(256) srs_powers_general_kernel(G1Affine *out, uint64_t start, uint64_t n, Fr tau) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr k = to_canonical(pow_u64(tau, start + i));
    G1Affine g; g.x = from_u64<FqParams>(1); g.y = from_u64<FqParams>(2);
    G1Xyzz p = xyzz_identity();
    for (int bit = 253; bit >= 0; bit--) {
        p = xyzz_double(p);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzz_add_mixed(p, g, false);
    }
    const G1Affine a = xyzz_to_affine_dev(p);
    store_fp(&out[i].x, a.x);
    store_fp(&out[i].y, a.y);
}

}  // namespace plk

using namespace plk;

// the same for an arbitrary tau given as a field element (Montgomery, as everywhere at this boundary)
extern "C" int32_t plk_srs_generate_fr(plk_ctx *ctx, uint64_t n, uint64_t start, const plk_fr *tau) {
    if (!ctx || n == 0 || !tau) { set_error("plk_srs_generate_fr: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_generate_fr"));
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->srs_own.reserve(n * sizeof(G1Affine)));
    Fr t; memcpy(t.l, tau->l, 32);
    hipLaunchKernelGGL(srs_powers_general_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->srs_own.as<G1Affine>(), start, n, t);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->srs = ctx->srs_own.p;
    ctx->srs_n = n;
    srs_table_invalidate(ctx);
    return PLK_OK;
}

// Fills the context's resident SRS with tau^(start+i) * G, i < n, tau a small integer (42 for crs_42).
extern "C" int32_t plk_srs_generate(plk_ctx *ctx, uint64_t n, uint64_t start, uint32_t tau) {
    if (!ctx || n == 0 || tau < 2) { set_error("plk_srs_generate: bad argument"); return PLK_ERR_ARG; }
    PLK_TRY(srs_replace_guard(ctx, "plk_srs_generate"));
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_TRY(ctx->srs_own.reserve(n * sizeof(G1Affine)));
    uint64_t threads = (n + SRS_RUN - 1) / SRS_RUN;
    hipLaunchKernelGGL(srs_powers_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->srs_own.as<G1Affine>(), start, n, from_u64<FrParams>(tau), tau);
    PLK_HIP(hipGetLastError());
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    ctx->srs = ctx->srs_own.p;
    ctx->srs_n = n;
    srs_table_invalidate(ctx);
    return PLK_OK;
}

// copies n resident SRS points (from index `offset`) back to the host
extern "C" int32_t plk_srs_download(plk_ctx *ctx, uint64_t offset, uint64_t n, plk_g1_affine *out) {
    if (!ctx || !out) { set_error("plk_srs_download: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->srs || offset + n > ctx->srs_n) { set_error("plk_srs_download: range outside the resident SRS"); return PLK_ERR_SRS; }
    PLK_HIP(hipSetDevice(ctx->device));
    PLK_HIP(hipMemcpyAsync(out, (const char *)ctx->srs + offset * sizeof(G1Affine), n * sizeof(G1Affine), hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}
