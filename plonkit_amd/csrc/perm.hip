// Copy-constraint permutation of the setup, built on the device.
//
// The reference's setup() (bellman_ce better_cs, reached from SetupForProver::prepare_setup_for_prover, src/plonk.rs:97-119)
// links all occurrences of a variable into one cycle; the rule pinned by the golden vk.bin is: occurrences
// in row-major order (gate by gate, a -> d inside a gate), each mapped to the NEXT one, the last to the first; the dummy
// variable (id 0) and the padding rows keep the identity.  On the host that is a counting sort over 4N (variable, position)
// pairs — 36 ms of single-threaded random access at 2^20 gates, plus a 16 MB upload.  Here the 4 x N variable-index table
// that the prover needs on the device anyway is the input: one STABLE radix sort by variable id (the positions are generated
// in row-major order, so equal keys stay in that order) and one pass that writes every occurrence's successor.
//
// The sort is a plain least-significant-digit radix sort over 8-bit digits, hand-written since round 3 (rounds 1-2 called
// rocPRIM's device-wide sort here — the one library primitive on the device side): per pass a histogram per 4096-element
// tile, an exclusive scan of the (digit, tile) table in digit-major order, and a scatter in which ONE wave walks its tile in
// index order, 64 elements at a time, ranking equal digits by eight ballots — stable by construction.  Setup path, 4N <= 2^28
// pairs of (u32 key, u32 position); 3 passes for the 2^20-gate circuits (21-bit variable ids): ~0.3 ms.
#include "ctx.h"
#include "poly.h"
#include <cstring>

namespace plk {

constexpr uint32_t RS_TILE = 4096;                 // elements per workgroup (one wave) and pass

// digit histogram of every tile: hist[digit * tiles + tile]
__global__ void __launch_bounds__(64) k_rs_hist(const uint32_t *keys, uint32_t n, uint32_t shift, uint32_t tiles, uint32_t *hist) {
    __shared__ uint32_t cnt[256];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) cnt[d] = 0;
    __syncthreads();
    const uint32_t lo = tile * RS_TILE, hi = lo + RS_TILE < n ? lo + RS_TILE : n;
    for (uint32_t i = lo + lane; i < hi; i += 64) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    for (uint32_t d = lane; d < 256; d += 64) hist[d * tiles + tile] = cnt[d];
}

// exclusive scan of the whole (digit, tile) table in place, digit-major: entry (d, t) becomes the number of elements with a
// smaller digit plus those with digit d in earlier tiles — the first output position of tile t's digit-d elements.
// Two levels since round 4 (rounds 1-3: ONE 1024-thread workgroup walking the table with a per-thread stride — 0.39 ms per pass
// at the 2^20 domain, the slowest kernel of the sort, and uncoalesced tens of megabytes at 2^26): blocks of 2048 words scanned
// in parallel with coalesced 16-byte accesses, their totals scanned by one workgroup, the offsets added back.
constexpr uint32_t RS_SCAN_BLOCK = 2048, RS_SCAN_THREADS = 256;     // 8 words per thread
__global__ void __launch_bounds__(RS_SCAN_THREADS) k_rs_scan_blocks(uint32_t *table, uint32_t total, uint32_t *block_sums) {
    __shared__ uint32_t sums[RS_SCAN_THREADS];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * RS_SCAN_BLOCK + tid * 8;
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = base + k < total ? table[base + k] : 0;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { const uint32_t c = v[k]; v[k] = s; s += c; }
    sums[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < RS_SCAN_THREADS; off <<= 1) {
        const uint32_t x = tid >= off ? sums[tid - off] : 0;
        __syncthreads();
        sums[tid] += x;
        __syncthreads();
    }
    const uint32_t before = tid ? sums[tid - 1] : 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (base + k < total) table[base + k] = v[k] + before;
    if (tid == RS_SCAN_THREADS - 1) block_sums[blockIdx.x] = sums[tid];
}
// exclusive scan of the block totals in place (one workgroup; up to 2^28 / 4096 * 256 / 2048 = 8192 of them)
__global__ void __launch_bounds__(1024) k_rs_scan_tops(uint32_t *block_sums, uint32_t nblocks) {
    __shared__ uint32_t sums[1024];
    const uint32_t tid = threadIdx.x, per = (nblocks + 1023) / 1024;
    uint32_t lo = tid * per, hi = lo + per < nblocks ? lo + per : nblocks;
    if (lo > nblocks) lo = nblocks;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; i++) s += block_sums[i];
    sums[tid] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t v = tid >= off ? sums[tid - off] : 0;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? sums[tid - 1] : 0;
    for (uint32_t i = lo; i < hi; i++) { const uint32_t c = block_sums[i]; block_sums[i] = run; run += c; }
}
__global__ void __launch_bounds__(RS_SCAN_THREADS) k_rs_scan_add(uint32_t *table, uint32_t total, const uint32_t *block_sums) {
    const uint32_t add = block_sums[blockIdx.x];
    if (!add) return;
    const uint32_t base = blockIdx.x * RS_SCAN_BLOCK + threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) if (base + k < total) table[base + k] += add;
}

// one wave per tile, elements in index order, 64 at a time: lanes holding the same digit find each other with eight
// ballots; an element's destination = the tile's base for its digit + the digit's running count in this tile + its rank
// among the equal-digit lanes below it
__global__ void __launch_bounds__(64) k_rs_scatter(const uint32_t *keys, const uint32_t *vals, uint32_t n, uint32_t shift, uint32_t tiles,
                                                   const uint32_t *offsets, uint32_t *out_keys, uint32_t *out_vals) {
    __shared__ uint32_t base[256];
    const uint32_t lane = threadIdx.x, tile = blockIdx.x;
    for (uint32_t d = lane; d < 256; d += 64) base[d] = offsets[d * tiles + tile];
    __syncthreads();
    const uint32_t lo = tile * RS_TILE, hi = lo + RS_TILE < n ? lo + RS_TILE : n;
    const uint64_t below = (1ull << lane) - 1;
    for (uint32_t r = lo; r < hi; r += 64) {
        const uint32_t i = r + lane;
        const bool live = i < hi;
        const uint32_t k = live ? keys[i] : 0, v = live ? vals[i] : 0, d = (k >> shift) & 255u;
        uint64_t peers = __ballot(live);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) {
            const uint64_t m = __ballot(live && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t dst = 0;
        if (live) dst = base[d] + (uint32_t)__popcll(peers & below);
        __syncthreads();                                       // (one wave: orders the LDS reads above before the updates below)
        if (live && (peers & below) == 0) base[d] += (uint32_t)__popcll(peers);      // the lowest lane of each group
        __syncthreads();
        if (live) { out_keys[dst] = k; out_vals[dst] = v; }
    }
}

// stable sort of n (key, value) pairs by the low `bits` bits of the key; the result lands in (keys_out, vals_out); the inputs are
// clobbered (ping-pong).  scratch: 256 * tiles words + one word per 2048 of them (block totals of the scan).
static int32_t radix_sort_pairs_stable(uint32_t *keys, uint32_t *vals, uint32_t *keys_out, uint32_t *vals_out, uint32_t n, uint32_t bits,
                                       uint32_t *scratch, hipStream_t st) {
    const uint32_t tiles = (n + RS_TILE - 1) / RS_TILE;
    uint32_t passes = (bits + 7) / 8;
    if (passes == 0) passes = 1;
    if (passes & 1) {                                           // an odd number of passes ends in the "out" pair if it starts in the "in" pair
    } else {                                                    // an even number: start from the out pair so that the last pass lands there
        PLK_HIP(hipMemcpyAsync(keys_out, keys, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        PLK_HIP(hipMemcpyAsync(vals_out, vals, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        uint32_t *t = keys; keys = keys_out; keys_out = t;
        t = vals; vals = vals_out; vals_out = t;
    }
    for (uint32_t p = 0; p < passes; p++) {
        hipLaunchKernelGGL(k_rs_hist, dim3(tiles), dim3(64), 0, st, (const uint32_t *)keys, n, 8 * p, tiles, scratch);
        {
            const uint32_t total = 256 * tiles, nblocks = (total + RS_SCAN_BLOCK - 1) / RS_SCAN_BLOCK;
            uint32_t *block_sums = scratch + total;
            hipLaunchKernelGGL(k_rs_scan_blocks, dim3(nblocks), dim3(RS_SCAN_THREADS), 0, st, scratch, total, block_sums);
            if (nblocks > 1) {
                hipLaunchKernelGGL(k_rs_scan_tops, dim3(1), dim3(1024), 0, st, block_sums, nblocks);
                hipLaunchKernelGGL(k_rs_scan_add, dim3(nblocks), dim3(RS_SCAN_THREADS), 0, st, scratch, total, (const uint32_t *)block_sums);
            }
        }
        hipLaunchKernelGGL(k_rs_scatter, dim3(tiles), dim3(64), 0, st, (const uint32_t *)keys, (const uint32_t *)vals, n, 8 * p, tiles,
                           (const uint32_t *)scratch, keys_out, vals_out);
        uint32_t *t = keys; keys = keys_out; keys_out = t;
        t = vals; vals = vals_out; vals_out = t;
    }
    PLK_HIP(hipGetLastError());
    return PLK_OK;
}

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
    const uint32_t tiles = (n4 + RS_TILE - 1) / RS_TILE;
    DevBuf buf;
    const size_t arr = ((size_t)n4 * 4 + 255) & ~(size_t)255;
    PLK_TRY(buf.reserve(4 * arr + (size_t)256 * tiles * 4 + ((size_t)256 * tiles / RS_SCAN_BLOCK + 2) * 4 + 256));
    uint32_t *keys = buf.as<uint32_t>(), *vals = keys + arr / 4, *skeys = vals + arr / 4, *svals = skeys + arr / 4;
    uint32_t *scratch = svals + arr / 4;
    const uint32_t blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(k_perm_keys, dim3(blocks), dim3(256), 0, st, vars[0], vars[1], vars[2], vars[3], n4, keys, vals);
    int32_t rc = radix_sort_pairs_stable(keys, vals, skeys, svals, n4, bits, scratch, st);
    hipError_t e = hipSuccess;
    if (rc == PLK_OK) {
        hipLaunchKernelGGL(k_perm_next, dim3(blocks), dim3(256), 0, st, (const uint32_t *)skeys, (const uint32_t *)svals, n4, n, idx);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);           // the scratch is released below
    buf.release();
    if (rc != PLK_OK) return rc;
    PLK_HIP(e);
    return PLK_OK;
}

}  // namespace plk
