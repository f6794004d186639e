// Inverse NTT over G1 — `plonkit dump-lagrange`:
//   Crs::<Bn256, CrsForLagrangeForm>::from_powers(&mono, n.next_power_of_two(), &Worker)
//   (src/plonk.rs:179-185; driven from src/bin/main.rs:360-381).
// in[j] = tau^j * G  ->  out[i] = L_i(tau) * G, the Lagrange-basis SRS of the size-N domain.
// Radix-2 DIT over group elements: a butterfly is (A, B) -> (A + w*B, A - w*B) where w*B is a full
// 254-bit scalar multiplication, so the kernel is bound by v_mad_u64_u32 issue (about 4000 modular
// multiplications per butterfly) and HBM traffic (128 B XYZZ per point per stage) is negligible;
// one lane per butterfly, XYZZ coordinates between stages, a single Fermat inversion per point at
// the end.  1/N is folded into the bit-reversing load pass.
#include "ctx.h"
#include "ec.cuh"
#include "ntt.h"

namespace plk {

__device__ __forceinline__ uint32_t brev32(uint32_t x, uint32_t bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// k * p, k canonical (non-Montgomery) 8x32 limbs, MSB-first double-and-add
__device__ __noinline__ G1Xyzz xyzz_mul_scalar(const G1Xyzz &p, const Fr &k) {
    G1Xyzz acc = xyzz_identity();
    if (is_inf(p)) return acc;
    int top = 253;
    while (top >= 0 && !((k.l[top >> 5] >> (top & 31)) & 1)) top--;
    for (int bit = top; bit >= 0; bit--) {
        acc = xyzz_double(acc);
        if ((k.l[bit >> 5] >> (bit & 31)) & 1) xyzz_add(acc, p);
    }
    return acc;
}

// pts[bitrev(i)] = n_inv * in[i]
__global__ void __launch_bounds__(256) g1ntt_load(G1Xyzz *pts, const G1Affine *in, uint32_t log_n, Fr n_inv_canon) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    G1Xyzz p = xyzz_from_affine(load_affine(in + i));
    store_xyzz(pts + brev32(i, log_n), xyzz_mul_scalar(p, n_inv_canon));
}

// one DIT stage with half-size h = 2^s
__global__ void __launch_bounds__(256) g1ntt_stage(G1Xyzz *pts, uint32_t log_n, uint32_t s, PowTable tw_inv) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (1u << (log_n - 1))) return;
    uint32_t h = 1u << s, jl = j & (h - 1);
    uint32_t i0 = ((j >> s) << (s + 1)) | jl, i1 = i0 + h;
    G1Xyzz a = load_xyzz(pts + i0), b = load_xyzz(pts + i1);
    if (jl) {
        // omega_N^-(jl * N / 2h)
        uint32_t e = (jl << (log_n - s - 1)) << (MAX_LOG_N - log_n);
        Fr w = mul(load_fp(tw_inv.lo + (e & (POW_TAB - 1))), load_fp(tw_inv.hi + (e >> POW_SPLIT)));
        b = xyzz_mul_scalar(b, to_canonical(w));
    }
    G1Xyzz lo = a, nb = xyzz_neg(b);
    xyzz_add(lo, b);
    xyzz_add(a, nb);
    store_xyzz(pts + i0, lo);
    store_xyzz(pts + i1, a);
}

__global__ void __launch_bounds__(256) g1ntt_to_affine(G1Affine *out, const G1Xyzz *pts, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Xyzz p = load_xyzz(pts + i);
    G1Affine a;
    if (is_inf(p)) { a.x = Fq::zero(); a.y = Fq::zero(); }
    else {
        Fq iv = inv(mul(p.zz, p.zzz));
        a.x = mul(p.x, mul(iv, p.zzz));
        a.y = mul(p.y, mul(iv, p.zz));
    }
    store_fp(&out[i].x, a.x);
    store_fp(&out[i].y, a.y);
}

int32_t g1_intt_dev(plk_ctx *ctx, const G1Affine *in, uint32_t log_n, G1Affine *out, hipStream_t st) {
    if (log_n > 26) { set_error("g1_intt: size exceeds 2^26"); return PLK_ERR_SIZE; }
    PLK_TRY(ntt_init_tables(ctx));
    const uint32_t n = 1u << log_n;
    PLK_TRY(ctx->slot[0].c.reserve((size_t)n * sizeof(G1Xyzz)));
    G1Xyzz *pts = ctx->slot[0].c.as<G1Xyzz>();
    Fr n_inv = to_canonical(ctx->n_inv[log_n]);
    hipLaunchKernelGGL(g1ntt_load, dim3((n + 255) / 256), dim3(256), 0, st, pts, in, log_n, n_inv);
    for (uint32_t s = 0; s < log_n; s++)
        hipLaunchKernelGGL(g1ntt_stage, dim3((n / 2 + 255) / 256), dim3(256), 0, st, pts, log_n, s, ctx->tw_inv);
    hipLaunchKernelGGL(g1ntt_to_affine, dim3((n + 255) / 256), dim3(256), 0, st, out, (const G1Xyzz *)pts, n);
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

}  // namespace plk

using namespace plk;

extern "C" int32_t plk_g1_intt(plk_ctx *ctx, const plk_g1_affine *in, uint32_t log_n, plk_g1_affine *out) {
    if (!ctx || !in || !out) { set_error("plk_g1_intt: bad argument"); return PLK_ERR_ARG; }
    if (log_n > 26) { set_error("g1_intt: size exceeds 2^26"); return PLK_ERR_SIZE; }
    PLK_HIP(hipSetDevice(ctx->device));
    const size_t bytes = sizeof(plk_g1_affine) << log_n;
    PLK_TRY(ctx->stage.reserve(2 * bytes));
    G1Affine *d_in = ctx->stage.as<G1Affine>(), *d_out = d_in + ((size_t)1 << log_n);
    PLK_HIP(hipMemcpyAsync(d_in, in, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLK_TRY(g1_intt_dev(ctx, d_in, log_n, d_out, ctx->stream));
    PLK_HIP(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLK_HIP(hipStreamSynchronize(ctx->stream));
    return PLK_OK;
}

// the same transform applied to the first 2^log_n points of the resident SRS; result left on the device
extern "C" int32_t plk_g1_intt_srs_dev(plk_ctx *ctx, uint32_t log_n, void *out_dev, void *stream) {
    if (!ctx || !out_dev) { set_error("plk_g1_intt_srs_dev: bad argument"); return PLK_ERR_ARG; }
    if (!ctx->srs || ctx->srs_n < ((uint64_t)1 << log_n)) { set_error("g1_intt: SRS too small"); return PLK_ERR_SRS; }
    PLK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    return g1_intt_dev(ctx, reinterpret_cast<const G1Affine *>(ctx->srs), log_n, reinterpret_cast<G1Affine *>(out_dev), s);
}
