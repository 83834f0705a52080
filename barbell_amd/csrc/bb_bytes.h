// bb_bytes.h — wave-level byte movers shared by the trim renderer and the FASTQ packer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// copies L bytes src -> dst with the 64 lanes of a wave: byte head up to a 16-byte boundary of dst,
// 16-byte chunks (unaligned loads, aligned stores), byte tail
// DEEP: four 1 KB rows of the wave in flight per iteration — pays for long copies in a light kernel (the FASTQ
// packer: +8 %), costs occupancy in a kernel with more live state (the trim renderer: -9 %)
template <bool DEEP = false>
static __device__ __forceinline__ void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t L, int lane) {
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > L) head = L;
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t body = (L - head) >> 4;
    const uint8_t* s = src + head;
    u32x4* d = (u32x4*)(dst + head);
    uint32_t c = (uint32_t)lane;
    if constexpr (DEEP) {
        for (; c + 192u < body; c += 256u) {
            u32x4 v0, v1, v2, v3;
            __builtin_memcpy(&v0, s + ((uint64_t)c << 4), 16);
            __builtin_memcpy(&v1, s + ((uint64_t)(c + 64u) << 4), 16);
            __builtin_memcpy(&v2, s + ((uint64_t)(c + 128u) << 4), 16);
            __builtin_memcpy(&v3, s + ((uint64_t)(c + 192u) << 4), 16);
            __builtin_nontemporal_store(v0, d + c);
            __builtin_nontemporal_store(v1, d + c + 64u);
            __builtin_nontemporal_store(v2, d + c + 128u);
            __builtin_nontemporal_store(v3, d + c + 192u);
        }
    }
    for (; c < body; c += 64u) {
        u32x4 v;
        __builtin_memcpy(&v, s + ((uint64_t)c << 4), 16);
        __builtin_nontemporal_store(v, d + c);
    }
    const uint32_t done = head + (body << 4);
    if (done + (uint32_t)lane < L) dst[done + lane] = src[done + lane];
}
