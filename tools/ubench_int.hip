// Instruction-rate microbenchmark for the integer / fp64 pipes that bound 256-bit modular
// arithmetic on gfx950.  Prints wave-instructions per cycle per SIMD-equivalent numbers used in
// DESIGN.md.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_int.hip -o ubench_int
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../plonkit_amd/csrc/field_dev.h"
using namespace plk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096, UNROLL = 16;

#define LOOP(ASMSTMT) for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { ASMSTMT; } }
// 4 independent chains each so latency is hidden even at low occupancy
#define K32(NAME, ASM)                                                                             \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                                           \
        uint32_t a = seed + threadIdx.x; uint32_t c0 = a; uint32_t c1 = a + 1; uint32_t c2 = a + 2; uint32_t c3 = a + 3; \
        LOOP(asm volatile(ASM : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a) : "vcc"))          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;                            \
    }
#define K64(NAME, ASM)                                                                             \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                                           \
        uint32_t a = seed + threadIdx.x; uint32_t b = seed * 3 + 1; uint64_t d = a;                \
        uint64_t c0 = a; uint64_t c1 = b; uint64_t c2 = a + b; uint64_t c3 = 7;                    \
        LOOP(asm volatile(ASM : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b), "v"(d) : "vcc")) \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3);                \
    }
K64(k_mad64, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3")
K64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %6\n v_lshl_add_u64 %1, %1, 0, %6\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %6")
K32(k_mullo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
K32(k_mulhi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4")
K32(k_mad24, "v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3")
K32(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4")
K32(k_addco, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc")
K32(k_add3, "v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0")
K32(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4")
K32(k_dot4, "v_dot4_u32_u8 %0, %4, %4, %0\n v_dot4_u32_u8 %1, %4, %4, %1\n v_dot4_u32_u8 %2, %4, %4, %2\n v_dot4_u32_u8 %3, %4, %4, %3")
__global__ void k_fma64(uint32_t *out, uint32_t seed) {
    double a = 1.0 + 1e-9 * (threadIdx.x + seed); double c0 = a; double c1 = a + 1; double c2 = a + 2; double c3 = a + 3;
    LOOP(asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3"
                      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a)))
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 + c1 + c2 + c3);
}

// full Montgomery products, CHAINS independent chains per thread
template <class F, int CHAINS>
__global__ void k_montmul(F *out, const F *in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x[CHAINS], y = load_fp(in + (i & 1023));
    for (int c = 0; c < CHAINS; c++) x[c] = load_fp(in + ((i + c + 1) & 1023));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = mul(x[c], y);
    }
    F acc = x[0];
    for (int c = 1; c < CHAINS; c++) acc = add(acc, x[c]);
    store_fp(out + i, acc);
}
template <class F>
__global__ void k_addsub(F *out, const F *in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x = load_fp(in + (i & 1023)), y = load_fp(in + ((i + 1) & 1023));
    for (int it = 0; it < iters; it++) { x = add(x, y); y = sub(y, x); }
    store_fp(out + i, add(x, y));
}

template <class K>
double time_ms(K launch, int reps = 5) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double ghz = prop.clockRate / 1e6;
    printf("device %s CUs=%d clock=%.2f GHz\n", prop.name, cus, ghz);
    uint32_t *out; CHECK(hipMalloc(&out, 256 * 1024 * 64 * 4));
    // 8 waves per SIMD: grid = CUs * 4 blocks of 512 threads
    const int blocks = cus * 4, threads = 512;
    const double waves = (double)blocks * threads / 64, per_wave_instr = (double)ITER * UNROLL * 4;
#define RUN(K) { double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 12345u); });      \
        double wi = waves * per_wave_instr;                                                                                 \
        printf("%-12s %8.3f ms  %7.2f Gwave-instr/s  = %.3f wave-instr/clk/SIMD (cycles/instr/SIMD %.2f)\n", #K, ms,      \
               wi / ms / 1e6, wi / (ms * 1e-3) / (cus * 4.0 * ghz * 1e9), (cus * 4.0 * ghz * 1e9) / (wi / (ms * 1e-3))); }
    RUN(k_mad64) RUN(k_mullo) RUN(k_mulhi) RUN(k_mad24) RUN(k_mulhi24) RUN(k_addco) RUN(k_add3) RUN(k_lshladd64) RUN(k_mov) RUN(k_fma64) RUN(k_dot4)

    std::vector<uint32_t> h(1024 * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> ((i % 8 == 7) ? 3 : 0);
    Fr *in, *o2; CHECK(hipMalloc(&in, 1024 * 32)); CHECK(hipMalloc(&o2, (size_t)cus * 16 * 256 * 32));
    CHECK(hipMemcpy(in, h.data(), 1024 * 32, hipMemcpyHostToDevice));
    const int iters = 2048;
    for (int bpc : {1, 2, 4, 8}) {
        int nb = cus * bpc;
#define RUNM(CH) { double ms = time_ms([&] { hipLaunchKernelGGL((k_montmul<Fr, CH>), dim3(nb), dim3(256), 0, 0, o2, in, iters); }); \
            double muls = (double)nb * 256 * iters * CH;                                                                            \
            printf("montmul Fr chains=%d blocks/CU=%d: %8.3f ms  %8.2f Gmul/s\n", CH, bpc, ms, muls / ms / 1e6); }
        RUNM(1) RUNM(2) RUNM(4)
    }
    { int nb = cus * 8; double ms = time_ms([&] { hipLaunchKernelGGL((k_montmul<Fq, 2>), dim3(nb), dim3(256), 0, 0, (Fq *)o2, (Fq *)in, iters); });
      printf("montmul Fq chains=2 blocks/CU=8: %8.3f ms  %8.2f Gmul/s\n", ms, (double)nb * 256 * iters * 2 / ms / 1e6); }
    { int nb = cus * 8; double ms = time_ms([&] { hipLaunchKernelGGL((k_addsub<Fr>), dim3(nb), dim3(256), 0, 0, o2, in, iters); });
      printf("add+sub Fr blocks/CU=8: %8.3f ms  %8.2f Gop/s\n", ms, (double)nb * 256 * iters * 2 / ms / 1e6); }
    return 0;
}
