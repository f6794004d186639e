// micro-benchmark (round 6): does a wave with fewer ACTIVE lanes run a dependent chain of field products / full additions faster on gfx950?
// (the reduction trees of the short-commitment path are bound by the dependent-issue latency of ONE wave per SIMD: DESIGN.md 4.2a)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonkit_amd/csrc tools/ubench_lanes.hip -o tools/ubench_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec29_dev.h"
using namespace plk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void __launch_bounds__(256, 1) k_chain(uint32_t *out, int iters, unsigned active) {
    const unsigned lane = threadIdx.x & 63;
    if (lane >= active) return;
    FqW9 a, b;
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & M29; b.l[i] = (blockIdx.x * 97u + i * 7919u + 5) & M29; }
    for (int it = 0; it < iters; it++) a = mulw_os<FqW>(a, b);         // a dependent chain of products (operand scanning: the latency form)
    uint32_t x = 0; for (int i = 0; i < 9; i++) x ^= a.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void __launch_bounds__(256, 1) k_chain_ps(uint32_t *out, int iters, unsigned active) {
    const unsigned lane = threadIdx.x & 63;
    if (lane >= active) return;
    FqW9 a, b;
    for (int i = 0; i < 9; i++) { a.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & M29; b.l[i] = (blockIdx.x * 97u + i * 7919u + 5) & M29; }
    for (int it = 0; it < iters; it++) a = mulw<FqW>(a, b);            // product scanning
    uint32_t x = 0; for (int i = 0; i < 9; i++) x ^= a.l[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <class K> double time_ms(K launch, int reps = 3) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best; }
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    const int iters = 400;
    for (int bpc : {1, 2, 4}) for (unsigned active : {64u, 32u, 16u, 4u}) {
        const int nb = cus * bpc;
        const double os = time_ms([&] { hipLaunchKernelGGL(k_chain, dim3(nb), dim3(256), 0, 0, out, iters, active); });
        const double ps = time_ms([&] { hipLaunchKernelGGL(k_chain_ps, dim3(nb), dim3(256), 0, 0, out, iters, active); });
        printf("%d wave(s)/SIMD, %2u active lanes: mulw_os %6.3f us per product   mulw %6.3f us per product\n", bpc, active, os * 1e3 / iters, ps * 1e3 / iters);
    }
    return 0;
}
