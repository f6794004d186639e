// micro-benchmark: chain of FULL XYZZ additions on the 29-bit layer (the step of the MSM reduction kernels)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonkit_amd/csrc tools/ubench_add.hip -o tools/ubench_add
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec29_dev.h"
using namespace plk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_gen(XyzzW *pts, unsigned n) {   // valid points i*G in XYZZ (W domain) with non-trivial ZZ
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine g; g.x = from_u64<FqParams>(1); g.y = from_u64<FqParams>(2);
    AffW q; q.x = csub_p(w_from_s(unpack<FqW>(g.x))); q.y = csub_p(w_from_s(unpack<FqW>(g.y)));
    XyzzW p = xyzzw_identity();
    for (unsigned b = 0; b < 12; b++) { p = xyzzw_double(p); if (((i + 5) >> (11 - b)) & 1) xyzzw_add_mixed(p, q, false); }
    store_xyzzw(pts + i, p);
}
// MODE 0: load the operand, then add (no overlap).  MODE 1: operand of the next step loaded before the addition.
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_add(XyzzW *out, const XyzzW *pts, int iters, unsigned mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    XyzzW acc = xyzzw_identity();
    unsigned idx = (unsigned)i * 2654435761u;
    XyzzW nxt;
    if (MODE == 1) nxt = load_xyzzw(pts + (idx & mask));
    for (int it = 0; it < iters; it++) {
        XyzzW o;
        if (MODE == 1) { o = nxt; idx = idx * 1664525u + 1013904223u; nxt = load_xyzzw(pts + (idx & mask)); }
        else { idx = idx * 1664525u + 1013904223u; o = load_xyzzw(pts + (idx & mask)); }
        xyzzw_add(acc, o);
    }
    store_xyzzw(out + i, acc);
}
template <class K> double time_ms(K launch, int reps = 3) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best; }
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    const unsigned NP = 1u << 12;
    XyzzW *pts, *out; CHECK(hipMalloc(&pts, (size_t)NP * sizeof(XyzzW))); CHECK(hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(XyzzW)));
    hipLaunchKernelGGL(k_gen, dim3(NP / 256), dim3(256), 0, 0, pts, NP); CHECK(hipDeviceSynchronize());
    const int iters = 64;
    for (int bpc : {1, 2}) {
        int nb = cus * bpc;
        { double ms = time_ms([&] { hipLaunchKernelGGL((k_add<0>), dim3(nb), dim3(256), 0, 0, out, pts, iters, NP - 1); });
          printf("full add, load-then-add, %d wave(s)/SIMD: %8.3f ms  %6.2f us per step  %7.3f G add/s\n", bpc, ms, ms * 1e3 / iters, (double)nb * 256 * iters / ms / 1e6); }
        { double ms = time_ms([&] { hipLaunchKernelGGL((k_add<1>), dim3(nb), dim3(256), 0, 0, out, pts, iters, NP - 1); });
          printf("full add, prefetched,    %d wave(s)/SIMD: %8.3f ms  %6.2f us per step  %7.3f G add/s\n", bpc, ms, ms * 1e3 / iters, (double)nb * 256 * iters / ms / 1e6); }
    }
    return 0;
}
