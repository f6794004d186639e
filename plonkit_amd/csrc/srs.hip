// SRS generation on the GPU: Crs::<Bn256, CrsForMonomialForm>::crs_42(size, &Worker)
// (src/plonk.rs:30-48 `gen_key_monomial_form`; `plonkit setup`, src/bin/main.rs:334-343).
// g1[i] = tau^i * G with the reference's insecure tau = 42 (SURVEY.md A.1 [derived]) — N fixed-base
// scalar multiplications.  Each lane owns a run of consecutive powers: one double-and-add to reach
// tau^(start) * G, then "multiply by tau" steps.  Since late round 6 the run's points leave as XYZZ and a second kernel
// brings eight of them to affine with ONE Fermat inversion (Montgomery's trick on ZZZ; 1 / Z = ZZ / ZZZ) instead of one
// inversion per point, and the ladder and the run are computed on the 9 x 29-bit lazy layer: 13.4 -> ~3 ms at 2^20 points.
#include "ctx.h"
#include "ec_dev.h"
#include "ec29_dev.h"
#include <cstring>
#include <cstdlib>

namespace plk {

constexpr uint32_t SRS_RUN = 16;                 // the direct kernel (used when the commitment scratch is busy)
constexpr uint32_t SRS_RUN_XYZZ = 16, SRS_NORM_K = 8;
constexpr uint64_t SRS_CHUNK = 1ull << 22;       // points per pass through the XYZZ scratch (512 MiB)

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

// the same runs on the 9 x 29-bit lazy layer (ec29_dev.h), points left in XYZZ (external form) for srs_to_affine_kernel.  At <= 2^20 points the launch is one wave
// per SIMD or less, so its duration is ONE lane's chain — the 254-bit ladder plus the run — and the lazy layer's products are half as long as the 32-bit layer's.
__global__ void __launch_bounds__(256) srs_powers_xyzz_kernel(G1Xyzz *out, uint64_t start, uint64_t n, Fr tau, uint32_t tau_small) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t i0 = t * SRS_RUN_XYZZ;
    if (i0 >= n) return;
    Fr k = to_canonical(pow_u64(tau, start + i0));
    AffW g;
    g.x = csub_p(w_from_s(unpack<FqW>(from_u64<FqParams>(1)))); g.y = csub_p(w_from_s(unpack<FqW>(from_u64<FqParams>(2))));
    XyzzW p = xyzzw_identity();
    for (int bit = 253; bit >= 0; bit--) {
        p = xyzzw_double(p);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzzw_add_mixed(p, g, false);
    }
    const int top = 31 - __clz((int)tau_small);                  // tau_small >= 2
    for (uint32_t j = 0; j < SRS_RUN_XYZZ && i0 + j < n; j++) {
        store_xyzz(out + i0 + j, xyzzw_export(p));
        XyzzW acc = p;                                            // p <- tau_small * p, MSB first
        for (int b = top - 1; b >= 0; b--) { acc = xyzzw_double(acc); if ((tau_small >> b) & 1) xyzzw_add(acc, p); }
        p = acc;
    }
}
// XYZZ -> affine, one inversion per SRS_NORM_K points; the second walk reads the points again
__global__ void __launch_bounds__(256) srs_to_affine_kernel(G1Affine *out, const G1Xyzz *in, uint64_t n) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lo = t * SRS_NORM_K, hi = lo + SRS_NORM_K < n ? lo + SRS_NORM_K : n;
    if (lo >= n) return;
    Fq prefix[SRS_NORM_K];
    Fq acc = Fq::one();
#pragma unroll
    for (uint32_t j = 0; j < SRS_NORM_K; j++) {
        if (lo + j < hi) { const Fq z = load_fp(&in[lo + j].zzz); if (!z.is_zero()) acc = mul(acc, z); }
        prefix[j] = acc;
    }
    Fq inv_acc = inv(acc);
#pragma unroll
    for (uint32_t jj = 0; jj < SRS_NORM_K; jj++) {
        const uint32_t j = SRS_NORM_K - 1 - jj;
        if (lo + j >= hi) continue;
        const G1Xyzz q = load_xyzz(in + lo + j);
        G1Affine a; a.x = Fq::zero(); a.y = Fq::zero();
        if (!is_inf(q)) {
            const Fq zi = j ? mul(inv_acc, prefix[j - 1]) : inv_acc;  // 1 / ZZZ_j
            inv_acc = mul(inv_acc, q.zzz);
            const Fq iz = mul(q.zz, zi), izz = mul(iz, iz);       // 1 / Z, 1 / ZZ
            a.x = mul(q.x, izz);
            a.y = mul(q.y, zi);
        }
        store_fp(&out[lo + j].x, a.x);
        store_fp(&out[lo + j].y, a.y);
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
    // the XYZZ scratch is the first commitment slot's: not while a commitment is in flight (PLK_SRS_DIRECT=1: the one-kernel path, A/B knob)
    static const bool direct_env = getenv("PLK_SRS_DIRECT") != nullptr;
    if (!direct_env && ctx->msm_enq == ctx->msm_fin) {
        const uint64_t chunk = n < SRS_CHUNK ? n : SRS_CHUNK;
        PLK_TRY(ctx->slot[0].e.reserve(chunk * sizeof(G1Xyzz)));
        G1Xyzz *tmp = ctx->slot[0].e.as<G1Xyzz>();
        for (uint64_t off = 0; off < n; off += chunk) {
            const uint64_t len = n - off < chunk ? n - off : chunk;
            const uint64_t threads = (len + SRS_RUN_XYZZ - 1) / SRS_RUN_XYZZ, nthreads = (len + SRS_NORM_K - 1) / SRS_NORM_K;
            hipLaunchKernelGGL(srs_powers_xyzz_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream, tmp, start + off, len, from_u64<FrParams>(tau), tau);
            hipLaunchKernelGGL(srs_to_affine_kernel, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream, ctx->srs_own.as<G1Affine>() + off, (const G1Xyzz *)tmp, len);
        }
    } else {
        uint64_t threads = (n + SRS_RUN - 1) / SRS_RUN;
        hipLaunchKernelGGL(srs_powers_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, ctx->stream,
                           ctx->srs_own.as<G1Affine>(), start, n, from_u64<FrParams>(tau), tau);
    }
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
