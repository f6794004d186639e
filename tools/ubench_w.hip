// throughput of the 9x29-bit field/EC layer: mulw chains and the mixed-addition loop in isolation
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../plonkit_amd/csrc/ec29_dev.h"
using namespace plk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int CHAINS>
__global__ void __launch_bounds__(256) k_mulw(Fq *out, const Fq *in, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    FqW9 x[CHAINS], y = unpack<FqW>(load_fp(in + (i & 1023)));
    for (int c = 0; c < CHAINS; c++) x[c] = unpack<FqW>(load_fp(in + ((i + c + 1) & 1023)));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = mulw(x[c], y);
    }
    FqW9 acc = x[0];
    for (int c = 1; c < CHAINS; c++) acc = addn(acc, x[c]);
    store_fp(out + i, pack<FqParams>(csub_p(mulw(acc, w_one<FqW>()))));
}
// the accumulate inner loop without the sort: every lane adds `iters` points (gathered with a stride) into one accumulator
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_madd(XyzzW *out, const G1Affine *pts, int iters, unsigned mask) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    XyzzW acc = xyzzw_identity();
    unsigned idx = (unsigned)i * 2654435761u;
    G1Affine pt = load_affine(pts + (idx & mask));
    for (int it = 0; it < iters; it++) {
        AffW cur; cur.x = unpack<FqW>(pt.x); cur.y = unpack<FqW>(pt.y);
        idx = idx * 1664525u + 1013904223u;
        pt = load_affine(pts + (idx & mask));
        xyzzw_add_mixed(acc, cur, (idx >> 30) & 1);
    }
    store_xyzzw(out + i, acc);
}
template <class K> double time_ms(K launch, int reps = 3) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize()); float best = 1e30f;
    for (int r = 0; r < reps; r++) { CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best; }
__global__ void k_gen(G1Affine *pts, unsigned n) {   // valid curve points: multiples of G via repeated mixed add (old layer), converted to the W domain
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine g; g.x = from_u64<FqParams>(1); g.y = from_u64<FqParams>(2);
    G1Xyzz p = xyzz_from_affine(g);
    for (unsigned b = 0; b < 20; b++) { p = xyzz_double(p); if ((i >> b) & 1) xyzz_add_mixed(p, g, false); }
    Fq iv = inv(mul(p.zz, p.zzz));
    Fq ax = mul(p.x, mul(iv, p.zzz)), ay = mul(p.y, mul(iv, p.zz));
    store_fp(&pts[i].x, pack<FqParams>(csub_p(w_from_s(unpack<FqW>(ax)))));
    store_fp(&pts[i].y, pack<FqParams>(csub_p(w_from_s(unpack<FqW>(ay)))));
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    std::vector<uint32_t> h(1024 * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) >> ((i % 8 == 7) ? 4 : 0);
    Fq *in, *o2; CHECK(hipMalloc(&in, 1024 * 32)); CHECK(hipMalloc(&o2, (size_t)cus * 16 * 256 * 32));
    CHECK(hipMemcpy(in, h.data(), 1024 * 32, hipMemcpyHostToDevice));
    const int iters = 2048;
    for (int bpc : {1, 2, 4, 8}) {
        int nb = cus * bpc;
#define RUNM(CH) { double ms = time_ms([&] { hipLaunchKernelGGL((k_mulw<CH>), dim3(nb), dim3(256), 0, 0, o2, in, iters); }); \
            printf("mulw chains=%d blocks/CU=%d: %8.3f ms  %8.2f Gmul/s\n", CH, bpc, ms, (double)nb * 256 * iters * CH / ms / 1e6); }
        RUNM(1) RUNM(2)
    }
    const unsigned NP = 1u << 20;
    G1Affine *pts; CHECK(hipMalloc(&pts, (size_t)NP * 64));
    hipLaunchKernelGGL(k_gen, dim3(NP / 256), dim3(256), 0, 0, pts, NP); CHECK(hipDeviceSynchronize());
    XyzzW *acc; CHECK(hipMalloc(&acc, (size_t)cus * 8 * 256 * sizeof(XyzzW)));
    const int mi = 256;
    for (int bpc : {1, 2, 3, 4}) {
        int nb = cus * bpc;
        { double ms = time_ms([&] { hipLaunchKernelGGL((k_madd<2>), dim3(nb), dim3(256), 0, 0, acc, pts, mi, NP - 1); });
          printf("madd minw=2 blocks/CU=%d: %8.3f ms  %8.3f Gmadd/s\n", bpc, ms, (double)nb * 256 * mi / ms / 1e6); }
        { double ms = time_ms([&] { hipLaunchKernelGGL((k_madd<4>), dim3(nb), dim3(256), 0, 0, acc, pts, mi, NP - 1); });
          printf("madd minw=4 blocks/CU=%d: %8.3f ms  %8.3f Gmadd/s\n", bpc, ms, (double)nb * 256 * mi / ms / 1e6); }
    }
    return 0;
}
