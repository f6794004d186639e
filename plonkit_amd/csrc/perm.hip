// Copy-constraint permutation of the setup, built on the device.
//
// The reference's setup() (bellman_ce better_cs, reached from SetupForProver::prepare_setup_for_prover, src/plonk.rs:97-119)
// links all occurrences of a variable into one cycle; the rule pinned by the golden vk.bin is: occurrences
// in row-major order (gate by gate, a -> d inside a gate), each mapped to the NEXT one, the last to the first; the dummy
// variable (id 0) and the padding rows keep the identity.  On the host that is a counting sort over 4N (variable, position)
// pairs — 36 ms of single-threaded random access at 2^20 gates, plus a 16 MB upload.  Here the 4 x N variable-index table
// that the prover needs on the device anyway is the input: one stable radix sort by variable id (rocPRIM's device-wide
// radix sort — a library primitive; the positions are generated in row-major order, so equal keys stay in that order) and
// one pass that writes every occurrence's successor.
#include "ctx.h"
#include "poly.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

namespace plk {

__global__ void __launch_bounds__(256) k_perm_keys(const uint32_t *v0, const uint32_t *v1, const uint32_t *v2, const uint32_t *v3, uint32_t n4, uint32_t *keys, uint32_t *vals) {
    const uint32_t pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= n4) return;
    const uint32_t col = pos & 3, row = pos >> 2;
    const uint32_t *v = col == 0 ? v0 : (col == 1 ? v1 : (col == 2 ? v2 : v3));
    keys[pos] = v[row];
    vals[pos] = pos;
}

// packed index of a position: column << 30 | row  (what k_sigma_from_index consumes)
__device__ __forceinline__ uint32_t packed_of(uint32_t pos) { return ((pos & 3) << 30) | (pos >> 2); }

__global__ void __launch_bounds__(256) k_perm_next(const uint32_t *skeys, const uint32_t *svals, uint32_t n4, uint32_t n, uint32_t *idx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint32_t v = skeys[i], pos = svals[i];
    uint32_t next = pos;                                          // dummy variable / padding: identity
    if (v != 0) {
        if (i + 1 < n4 && skeys[i + 1] == v) next = svals[i + 1];
        else {                                                    // last occurrence: back to the first (lower bound of v)
            uint32_t lo = 0, hi = i;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skeys[mid] < v) lo = mid + 1; else hi = mid; }
            next = svals[lo];
        }
    }
    idx[(size_t)(pos & 3) * n + (pos >> 2)] = packed_of(next);
}

// idx: 4 x n packed successors (column-major: idx[col * n + row]); vars: the four columns of the wire -> variable table
int32_t build_permutation_index(plk_ctx *ctx, const uint32_t *const vars[4], uint32_t n, uint64_t num_vars, uint32_t *idx, hipStream_t st) {
    const uint32_t n4 = 4 * n;
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < num_vars) bits++;
    size_t tmp_bytes = 0;
    uint32_t *nul = nullptr;
    PLK_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, nul, nul, nul, nul, (size_t)n4, 0u, bits, st));
    DevBuf buf;
    const size_t arr = ((size_t)n4 * 4 + 255) & ~(size_t)255;
    PLK_TRY(buf.reserve(4 * arr + tmp_bytes + 256));
    uint32_t *keys = buf.as<uint32_t>(), *vals = keys + arr / 4, *skeys = vals + arr / 4, *svals = skeys + arr / 4;
    void *tmp = reinterpret_cast<char *>(buf.p) + 4 * arr;
    const uint32_t blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_perm_keys, dim3(blocks), dim3(256), 0, st, vars[0], vars[1], vars[2], vars[3], n4, keys, vals);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, skeys, vals, svals, (size_t)n4, 0u, bits, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_perm_next, dim3(blocks), dim3(256), 0, st, (const uint32_t *)skeys, (const uint32_t *)svals, n4, n, idx);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // the scratch is released below
    buf.release();
    PLK_HIP(e);
    return PLK_OK;
}

}  // namespace plk
