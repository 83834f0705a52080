// bb_bytes.h — wave-level byte movers shared by the trim renderer and the FASTQ packer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// copies L bytes src -> dst with the 64 lanes of a wave: byte head up to a 16-byte boundary of dst,
// 16-byte chunks (unaligned loads, aligned stores), byte tail.  (Keeping four 1 KB rows of the wave in flight
// per iteration was measured and is slower: 14.3 vs 13.4 ms for the FASTQ packer's 16 GB block.)
static __device__ __forceinline__ void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, int lane) {
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > L) head = L;
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t body = (L - head) >> 4;
    const uint8_t* s = src + head;
    u32x4* d = (u32x4*)(dst + head);
    uint32_t c = (uint32_t)lane;
    for (; c < body; c += 64u) {
        u32x4 v;
        __builtin_memcpy(&v, s + ((uint64_t)c << 4), 16);
        __builtin_nontemporal_store(v, d + c);
    }
    const uint32_t done = head + (body << 4);
    if (done + (uint32_t)lane < L) dst[done + lane] = src[done + lane];
}
