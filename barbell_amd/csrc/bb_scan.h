// bb_scan.h — the scans and the stable radix sort the steps around annotate need (FASTQ ingest: record offsets; TSV renderer:
// line positions; trim/split: records grouped by output label).  Hand-written for wave64: one launch sequence per call, no
// temporary-storage protocol, no library underneath.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- 64-bit exclusive scan of u32 values, 1024 per block; out has n+1 entries (out[n] = total) -------------------------
static __global__ __launch_bounds__(256) void k_scan64_block(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n, uint64_t* __restrict__ sums) {
    __shared__ uint64_t s_w[4];
    const uint32_t b0 = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint64_t t = 0, pre[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { pre[i] = t; if (b0 + i < n) t += in[b0 + i]; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint64_t wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += s_w[i];
    const uint64_t excl = wbase + inc - t;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (b0 + i < n) out[b0 + i] = excl + pre[i];
    if (threadIdx.x == 255) sums[blockIdx.x] = wbase + inc;
}
static __global__ __launch_bounds__(64) void k_scan64_sums(uint64_t* __restrict__ sums, uint32_t nb, uint64_t* __restrict__ total) {
    uint64_t carry = 0;
    const int lane = threadIdx.x;
    for (uint32_t b = 0; b < nb; b += 64) {
        const uint64_t x = b + lane < nb ? sums[b + lane] : 0ull;
        uint64_t inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        if (b + lane < nb) sums[b + lane] = carry + inc - x;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[0] = carry;
}
static __global__ __launch_bounds__(256) void k_scan64_add(uint64_t* __restrict__ out, uint32_t n, const uint64_t* __restrict__ sums, const uint64_t* __restrict__ total) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] += sums[i >> 10];
    if (i == 0) out[n] = total[0];
}
// d_sums: (n + 1023) / 1024 + 1 entries of scratch; d_total: one u64
static inline hipError_t bb_scan64(hipStream_t st, const uint32_t* in, uint64_t* out, uint32_t n, uint64_t* d_sums, uint64_t* d_total) {
    const uint32_t nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_scan64_block, dim3(nb), dim3(256), 0, st, in, out, n, d_sums);
    hipLaunchKernelGGL(k_scan64_sums, dim3(1), dim3(64), 0, st, d_sums, nb, d_total);
    hipLaunchKernelGGL(k_scan64_add, dim3((n + 255) / 256), dim3(256), 0, st, out, n, (const uint64_t*)d_sums, (const uint64_t*)d_total);
    return hipGetLastError();
}

// ---- u32 exclusive scan, 2048 per block ------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_scan32_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint64_t n, uint32_t* __restrict__ sums) {
    __shared__ uint32_t s_w[4];
    const uint64_t b0 = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    uint32_t v[8], t = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = b0 + i < n ? in[b0 + i] : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint32_t x = v[i]; v[i] = t; t += x; }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int i = 0; i < wv; ++i) wbase += s_w[i];
    const uint32_t excl = wbase + inc - t;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (b0 + i < n) out[b0 + i] = v[i] + excl;
    if (threadIdx.x == 255) sums[blockIdx.x] = wbase + inc;
}
// single-wave scan of the block sums; T = uint32_t or uint64_t.  total[0] = grand total.
template <typename T>
static __global__ __launch_bounds__(64) void k_scan_sums_t(T* __restrict__ sums, uint32_t nb, T* __restrict__ total) {
    T carry = 0;
    const int lane = threadIdx.x;
    for (uint32_t b = 0; b < nb; b += 64) {
        const T x = b + lane < nb ? sums[b + lane] : (T)0;
        T inc = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const T y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        if (b + lane < nb) sums[b + lane] = carry + inc - x;
        carry += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[0] = carry;
}
static __global__ __launch_bounds__(256) void k_scan32_add(uint32_t* __restrict__ out, uint64_t n, const uint32_t* __restrict__ sums) {
    const uint64_t b0 = (uint64_t)blockIdx.x * 2048u + (uint64_t)threadIdx.x * 8u;
    const uint32_t a = sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; ++i) if (b0 + i < n) out[b0 + i] += a;
}
// d_sums: (n + 2047) / 2048 + 1 entries of scratch; d_total: one u32 (may be null-free scratch)
static inline hipError_t bb_scan32(hipStream_t st, const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* d_sums, uint32_t* d_total) {
    const uint32_t nb = (uint32_t)((n + 2047) / 2048);
    hipLaunchKernelGGL(k_scan32_block, dim3(nb), dim3(256), 0, st, in, out, n, d_sums);
    hipLaunchKernelGGL(k_scan_sums_t<uint32_t>, dim3(1), dim3(64), 0, st, d_sums, nb, d_total);
    hipLaunchKernelGGL(k_scan32_add, dim3(nb), dim3(256), 0, st, out, n, (const uint32_t*)d_sums);
    return hipGetLastError();
}

// ---- stable LSD radix sort of (u32 key, u32 value) pairs, 8 bits per pass ------------------------------------------------
// One wave per tile of BB_RS_TILE consecutive pairs.  Pass = digit histogram per tile (LDS atomics) -> exclusive scan of
// hist[digit][tile] (digit-major: the scan IS the global offset of every (digit, tile)) -> scatter: the wave walks its tile
// 64 pairs a round, in order; the lanes holding the same digit find each other with eight ballots (one per digit bit), a lane's
// rank among them is a popcount, their leader advances the tile's running offset of that digit in LDS.  Order inside a digit is
// input order at every step, so the sort is stable.  The caller passes only the digit positions its keys can occupy.
#define BB_RS_TILE 1024u
static __global__ __launch_bounds__(64) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t ntiles, uint32_t* __restrict__ hist) {
    __shared__ uint32_t s_h[256];
    const uint32_t lane = threadIdx.x, t0 = blockIdx.x * BB_RS_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_h[lane + 64 * i] = 0u;
    __syncthreads();
    for (uint32_t r = 0; r < BB_RS_TILE; r += 64u) {
        const uint32_t i = t0 + r + lane;
        if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) hist[(size_t)(lane + 64 * i) * ntiles + blockIdx.x] = s_h[lane + 64 * i];
}
static __global__ __launch_bounds__(64) void k_rs_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t shift,
                                                          uint32_t ntiles, const uint32_t* __restrict__ offs, uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t s_b[256];
    const uint32_t lane = threadIdx.x, t0 = blockIdx.x * BB_RS_TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_b[lane + 64 * i] = offs[(size_t)(lane + 64 * i) * ntiles + blockIdx.x];
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < BB_RS_TILE; r += 64u) {
        const uint32_t i = t0 + r + lane;
        const bool on = i < n;
        const uint32_t k = on ? keys[i] : 0u, v = on ? vals[i] : 0u, d = (k >> shift) & 0xFFu;
        unsigned long long peers = __ballot(on);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below), cnt = (uint32_t)__popcll(peers);
        uint32_t base = 0u;
        if (on && rank == 0u) { base = s_b[d]; s_b[d] = base + cnt; }
        base = (uint32_t)__shfl((int)base, on ? (int)__ffsll((long long)peers) - 1 : 0, 64);
        if (on) { keys_out[base + rank] = k; vals_out[base + rank] = v; }
        __syncthreads();
    }
}
// Sorts n pairs by the key bits covered by `shifts` (each entry = the low bit of an 8-bit digit, ascending).  k0/v0 hold the
// input, k1/v1 are scratch of the same size; the result's location is returned through *k_res / *v_res.  d_hist:
// 256 * ntiles + 1 entries, d_sums: (256 * ntiles) / 2048 + 2 entries.
static inline hipError_t bb_radix_sort_pairs(hipStream_t st, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t n, const uint32_t* shifts,
                                             int n_shifts, uint32_t* d_hist, uint32_t* d_sums, uint32_t** k_res, uint32_t** v_res) {
    const uint32_t ntiles = (n + BB_RS_TILE - 1) / BB_RS_TILE;
    for (int p = 0; p < n_shifts; ++p) {
        hipLaunchKernelGGL(k_rs_hist, dim3(ntiles), dim3(64), 0, st, (const uint32_t*)k0, n, shifts[p], ntiles, d_hist);
        const hipError_t e = bb_scan32(st, d_hist, d_hist, (uint64_t)256 * ntiles, d_sums, d_hist + (size_t)256 * ntiles);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_rs_scatter, dim3(ntiles), dim3(64), 0, st, (const uint32_t*)k0, (const uint32_t*)v0, n, shifts[p], ntiles,
                           (const uint32_t*)d_hist, k1, v1);
        uint32_t* t = k0; k0 = k1; k1 = t;
        t = v0; v0 = v1; v1 = t;
    }
    *k_res = k0; *v_res = v0;
    return hipGetLastError();
}
