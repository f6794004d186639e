// second instruction-rate probe: independent chains, VOP2 vs VOP3 forms, 64-bit shifts, carries
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4096, UNROLL = 16;
#define LOOP(ASMSTMT) for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int u = 0; u < UNROLL; u++) { ASMSTMT; } }
// 8 independent 32-bit chains c0..c7, sources a, b
#define K32(NAME, ASM)                                                                             \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                                           \
        uint32_t a = seed + threadIdx.x, b = seed * 7 + 3;                                         \
        uint32_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7; \
        LOOP(asm volatile(ASM : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23")) \
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;        \
    }
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
K32(k_add_u32,   "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8")
K32(k_and,       "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8")
K32(k_lshr,      "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7")
K32(k_addco_ind, "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n v_add_co_u32 %4, vcc, %4, %8\n v_add_co_u32 %5, vcc, %5, %8\n v_add_co_u32 %6, vcc, %6, %8\n v_add_co_u32 %7, vcc, %7, %8")
K32(k_addc_chain,"v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc")
K32(k_add3_ind,  "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9")
K32(k_alignbit,  "v_alignbit_b32 %0, %0, %8, 29\n v_alignbit_b32 %1, %1, %8, 29\n v_alignbit_b32 %2, %2, %8, 29\n v_alignbit_b32 %3, %3, %8, 29\n v_alignbit_b32 %4, %4, %8, 29\n v_alignbit_b32 %5, %5, %8, 29\n v_alignbit_b32 %6, %6, %8, 29\n v_alignbit_b32 %7, %7, %8, 29")
K32(k_and_or,    "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9")
K32(k_lshl_add,  "v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8")
K32(k_cndmask,   "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
K32(k_mullo_ind, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8")
K32(k_mul_u24,   "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8")
K32(k_subrev,    "v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %8, %4\n v_subrev_u32 %5, %8, %5\n v_subrev_u32 %6, %8, %6\n v_subrev_u32 %7, %8, %7")
#define K64(NAME, ASM)                                                                             \
    __global__ void NAME(uint32_t *out, uint32_t seed) {                                           \
        uint32_t a = seed + threadIdx.x, b = seed * 7 + 3; uint64_t d = a;                         \
        uint64_t c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3, c4 = a + 4, c5 = a + 5, c6 = a + 6, c7 = a + 7; \
        LOOP(asm volatile(ASM : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b), "v"(d) : "vcc", "s20", "s21")) \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7); \
    }
K64(k_mad64_ind, "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7")
K64(k_lshr64,    "v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7")
K64(k_add64,     "v_lshl_add_u64 %0, %0, 0, %10\n v_lshl_add_u64 %1, %1, 0, %10\n v_lshl_add_u64 %2, %2, 0, %10\n v_lshl_add_u64 %3, %3, 0, %10\n v_lshl_add_u64 %4, %4, 0, %10\n v_lshl_add_u64 %5, %5, 0, %10\n v_lshl_add_u64 %6, %6, 0, %10\n v_lshl_add_u64 %7, %7, 0, %10")
template <class K> double time_ms(K launch, int reps = 5) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best; }
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double ghz = prop.clockRate / 1e6;
    uint32_t *out; CHECK(hipMalloc(&out, 256 * 1024 * 64 * 4));
    const int blocks = cus * 4, threads = 512; const double waves = (double)blocks * threads / 64, per_wave = (double)ITER * UNROLL * 8;
#define RUN(K) { double ms = time_ms([&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, 12345u); }); double wi = waves * per_wave; \
        printf("%-14s %8.3f ms  cycles/wave-instr/SIMD (at %.1f GHz nominal) %.2f\n", #K, ms, ghz, (cus * 4.0 * ghz * 1e9) / (wi / (ms * 1e-3))); }
    RUN(k_add_u32) RUN(k_and) RUN(k_lshr) RUN(k_subrev) RUN(k_addco_ind) RUN(k_addc_chain) RUN(k_add3_ind) RUN(k_alignbit) RUN(k_and_or) RUN(k_lshl_add) RUN(k_cndmask)
    RUN(k_mullo_ind) RUN(k_mul_u24) RUN(k_mad64_ind) RUN(k_lshr64) RUN(k_add64)
    return 0; }
