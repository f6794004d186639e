// Device-side check of the DISTRIBUTED four-lane addition (plonkit_amd/csrc/ec29_quad_dev.h: xyzzw_add_dist, quad_distribute, quad_gather) against the
// lane-wise XYZZ addition of ec29_dev.h, which tests/host/field29_check.hip pins to the 8 x 32-bit layer and the GPU suite to the oracle.
// Every quad of the launch builds two points k_a * G and k_b * G (G = (1, 2), small random k: a double-and-add walk on the lane-wise formulas), adds them both
// ways and compares the results projectively; the special cases are planted by quad index: B = A (doubling), B = -A, A = infinity, B = infinity, both.
// Then a chain: the distributed sum is fed back 24 times (X <- X + B), as the reduction trees do, and compared with the lane-wise chain at the end.
// Built and run by tests/test_gpu_quad_add.py (-m gpu).  Prints "quad_add: <n> mismatches of <cases>".
#include "ec29_quad_dev.h"
#include <cstdio>
#include <hip/hip_runtime.h>
using namespace plk;

__device__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ XyzzW small_multiple(uint32_t k) {                      // k * G by double-and-add (lane-wise formulas), k != 0
    G1Affine g; g.x = Fq::one(); g.y = add(Fq::one(), Fq::one());
    AffW gw; gw.x = csub_p(w_from_s(unpack<FqW>(g.x))); gw.y = csub_p(w_from_s(unpack<FqW>(g.y)));
    XyzzW acc = xyzzw_identity();
    for (int i = 19; i >= 0; i--) {
        acc = xyzzw_double(acc);
        if ((k >> i) & 1) xyzzw_add_mixed(acc, gw, false);
    }
    return acc;
}
__device__ bool same_point(const XyzzW &p, const XyzzW &q) {      // projective equality: X1 ZZ2 = X2 ZZ1, Y1 ZZZ2 = Y2 ZZZ1
    if (is_inf(p) || is_inf(q)) return is_inf(p) && is_inf(q);
    return is_zero_mod_p(sub2(LM(p.x, q.zz), LM(q.x, p.zz))) && is_zero_mod_p(sub2(LM(p.y, q.zzz), LM(q.y, p.zzz)));
}

__global__ void __launch_bounds__(256) check(uint32_t seed, uint32_t *bad, uint32_t *special_seen) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, quad = tid >> 2, role = tid & 3;
    const uint32_t ka = (mix(seed + 2 * quad) & 0xfffffu) | 1u, kb = (mix(seed + 2 * quad + 1) & 0xfffffu) | 1u;
    XyzzW A = small_multiple(ka), B = small_multiple(kb);
    const uint32_t kind = quad & 7u;
    if (kind == 0) B = A;                                         // doubling through the rare branch
    if (kind == 1) { B = A; B.y = sub6(w_zero<FqW>(), A.y); }     // opposite points (6p - y: y < 6p by the bounds of ec29_dev.h) -> infinity
    if (kind == 2) A = xyzzw_identity();
    if (kind == 3) B = xyzzw_identity();
    if (kind == 4) { A = xyzzw_identity(); B = xyzzw_identity(); }
    XyzzW want = A;
    xyzzw_add(want, B);
    // every lane holds A and B in full: the distributed operands are one select away
    auto coord_of = [&](const XyzzW &p) { return wsel(role < 2, wsel(role == 0, p.x, p.y), wsel(role == 2, p.zz, p.zzz)); };
    const FqW9 got_d = xyzzw_add_dist(coord_of(A), coord_of(B), role);
    const XyzzW got = quad_gather(got_d);
    uint32_t wrong = same_point(got, want) ? 0u : 1u;
    if (kind == 1 && !is_inf(got)) wrong = 1;
    // quad_distribute: the point held by lane SRC must arrive as that lane's coordinates
    {
        XyzzW mine = role == 0 ? A : role == 1 ? B : role == 2 ? want : got;       // four different points in the four lanes
        const XyzzW back = quad_gather(quad_distribute<2>(mine, role));
        if (!same_point(back, want)) wrong = 1;
    }
    // a chain of 24 dependent additions, fed back in the distributed form (bounds of repeated use)
    if (kind >= 5) {
        FqW9 X = coord_of(A);
        XyzzW L = A;
        for (int i = 0; i < 24; i++) { X = xyzzw_add_dist(X, coord_of(B), role); xyzzw_add(L, B); }
        if (!same_point(quad_gather(X), L)) wrong = 1;
    }
    if (wrong && role == 0) atomicAdd(bad, 1u);
    if (role == 0 && kind < 5) atomicAdd(special_seen + kind, 1u);
}

int main() {
    uint32_t *bad, *seen;
    if (hipMalloc(&bad, 4) != hipSuccess || hipMalloc(&seen, 20) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemset(bad, 0, 4); (void)hipMemset(seen, 0, 20);
    const uint32_t blocks = 64, quads = blocks * 256 / 4;
    for (uint32_t round = 0; round < 4; round++) hipLaunchKernelGGL(check, dim3(blocks), dim3(256), 0, 0, 1234567u + 7919u * round, bad, seen);
    uint32_t h_bad = 0, h_seen[5];
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
    (void)hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h_seen, seen, 20, hipMemcpyDeviceToHost);
    printf("special cases seen: doubling %u opposite %u inf+B %u A+inf %u inf+inf %u\n", h_seen[0], h_seen[1], h_seen[2], h_seen[3], h_seen[4]);
    printf("quad_add: %u mismatches of %u\n", h_bad, 4 * quads);
    return h_bad ? 1 : 0;
}
