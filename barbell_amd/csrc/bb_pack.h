// bb_pack.h — two bases per byte (what BB_FASTQ_PACKED blocks and bb_annotate_batch_packed carry over PCIe): every kernel looks at a read
// character only through its IUPAC base set, a 4-bit code, so the host may keep exactly that.  byte = (code[2i] << 4) | (code[2i+1] ^ 0xA);
// an odd last base is paired with `next` (15 = none).  Shared by the C ABI's bb_pack_bases (bb_pack_abi.cpp) and the C++ host's block
// feeder (host/bb_host.cpp).  Host code only.
#pragma once
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>

#include <array>

static const uint8_t* base_code_table() {
    static const std::array<uint8_t, 256> T = []() {
        std::array<uint8_t, 256> t{};
        const char* letters = "ACGTURYSWKMBDHVN";
        const uint8_t codes[16] = {1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};   // bb_text_code (bb_common.h): A=1 C=2 G=4 T=8 and unions; X and non-letters 0
        for (int i = 0; i < 16; ++i) { t[(uint8_t)letters[i]] = codes[i]; t[(uint8_t)(letters[i] | 0x20)] = codes[i]; }
        return t;
    }();
    return T.data();
}
// 32 characters -> 16 packed bytes with AVX2 (the readers pack ~50 GB/s of sequence text at 12 M reads/s: a byte-wise table walk was the slowest
// stage of the whole FASTQ -> TSV pipeline).  Letters fold to upper case (c & 0xDF), 'A'..'Z' index two 16-entry vpshufb tables, anything
// else is 0; vpmaddubsw forms (c0 << 4) + (c1 ^ 0xA) per pair.  Returns false if some pair was (0, 0) (no packed form).
__attribute__((target("avx2"))) static bool pack32_avx2(uint8_t* out, const uint8_t* in) {
    const __m256i T0 = _mm256_setr_epi8(0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0, 0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0);   // @ A B C D E F G H I J K L M N O
    const __m256i T1 = _mm256_setr_epi8(0, 0, 5, 6, 8, 8, 7, 9, 0, 10, 0, 0, 0, 0, 0, 0, 0, 0, 5, 6, 8, 8, 7, 9, 0, 10, 0, 0, 0, 0, 0, 0);          // P Q R S T U V W X Y Z
    const __m256i x = _mm256_loadu_si256((const __m256i*)in);
    const __m256i v = _mm256_xor_si256(_mm256_and_si256(x, _mm256_set1_epi8((char)0xDF)), _mm256_set1_epi8(0x40));   // letters: 1..26
    const __m256i vm1 = _mm256_sub_epi8(v, _mm256_set1_epi8(1));
    const __m256i valid = _mm256_cmpeq_epi8(_mm256_min_epu8(vm1, _mm256_set1_epi8(25)), vm1);
    const __m256i r = _mm256_blendv_epi8(_mm256_shuffle_epi8(T0, v), _mm256_shuffle_epi8(T1, v), _mm256_slli_epi16(v, 3));
    const __m256i c = _mm256_and_si256(r, valid);
    const bool ok = _mm256_movemask_epi8(_mm256_cmpeq_epi16(c, _mm256_setzero_si256())) == 0;
    const __m256i m = _mm256_maddubs_epi16(_mm256_xor_si256(c, _mm256_set1_epi16(0x0A00)), _mm256_set1_epi16(0x0110));   // c0 * 16 + (c1 ^ 0xA) * 1
    const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(m, m), 0x08);
    _mm_storeu_si128((__m128i*)out, _mm256_castsi256_si128(pk));
    return ok;
}
static const bool g_have_avx2 = (__builtin_cpu_init(), __builtin_cpu_supports("avx2")) && !getenv("BARBELL_AMD_NO_AVX2");
// 64 characters -> 32 packed bytes with AVX-512 VBMI (Zen 4 / 5, Ice Lake and later): the character's low six bits index ONE 64-entry vpermb
// table (upper case 1..26, lower case 33..58), the mask keeps bytes 0x40..0x7F only; vpmaddubsw pairs, vpmovwb narrows.
__attribute__((target("avx512f,avx512bw,avx512vbmi"))) static bool pack64_avx512(uint8_t* out, const uint8_t* in) {
    alignas(64) static const uint8_t T[64] = {0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0, 0, 0, 5, 6, 8, 8, 7, 9, 0, 10, 0, 0, 0, 0, 0, 0,
                                              0, 1, 14, 2, 13, 0, 0, 4, 11, 0, 0, 12, 0, 3, 15, 0, 0, 0, 5, 6, 8, 8, 7, 9, 0, 10, 0, 0, 0, 0, 0, 0};
    const __m512i x = _mm512_loadu_si512((const void*)in);
    const __mmask64 letter = _mm512_cmpeq_epi8_mask(_mm512_and_si512(x, _mm512_set1_epi8((char)0xC0)), _mm512_set1_epi8(0x40));
    const __m512i c = _mm512_maskz_permutexvar_epi8(letter, x, _mm512_load_si512((const void*)T));   // vpermb uses the low six bits of each index byte
    const bool ok = _mm512_cmpeq_epi16_mask(c, _mm512_setzero_si512()) == 0;
    const __m512i m = _mm512_maddubs_epi16(_mm512_xor_si512(c, _mm512_set1_epi16(0x0A00)), _mm512_set1_epi16(0x0110));
    _mm256_storeu_si256((__m256i*)out, _mm512_cvtepi16_epi8(m));
    return ok;
}
static const bool g_have_avx512 = (__builtin_cpu_init(), __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512bw")) &&
                                  !getenv("BARBELL_AMD_NO_AVX512") && !getenv("BARBELL_AMD_NO_AVX2");
// the bases [b, e) of a sequence line, b at an even position of the line, as packed bytes; `next` pairs with a last unpaired base (15 = none)
static inline size_t pack_bases(uint8_t* out, const uint8_t* b, const uint8_t* e, uint8_t next, bool& unpackable) {
    const uint8_t* C = base_code_table();
    size_t d = 0;
    uint8_t z = 0xFF;   // AND of (c0 | c1) != 0 over the pairs, folded: becomes 0 if some pair was (0, 0)
    if (g_have_avx512) {
        bool ok = true;
        for (; b + 64 <= e; b += 64, d += 32) ok &= pack64_avx512(out + d, b);
        if (!ok) z = 0;
    }
    if (g_have_avx2) {
        bool ok = true;
        for (; b + 32 <= e; b += 32, d += 16) ok &= pack32_avx2(out + d, b);
        if (!ok) z = 0;
    }
    for (; b + 1 < e; b += 2) {
        const uint8_t c0 = C[b[0]], c1 = C[b[1]];
        z &= (uint8_t)((c0 | c1) ? 0xFF : 0);
        out[d++] = (uint8_t)((c0 << 4) | (c1 ^ 0xA));
    }
    if (b < e) {
        const uint8_t c0 = C[b[0]];
        z &= (uint8_t)((c0 | next) ? 0xFF : 0);
        out[d++] = (uint8_t)((c0 << 4) | (next ^ 0xA));
    }
    if (!z) unpackable = true;
    return d;
}

