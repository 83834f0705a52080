// valu_ceiling.hip — measures what a gfx950 SIMD can issue per second for the instruction classes the
// annotate kernels are made of (VERDICT r01 "Next round" #3): dependent chains (latency) and 8 independent
// chains (throughput) of each class at 1/2/3/4/6/8 waves per SIMD.  Prints one JSON object.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_ceiling tools/valu_ceiling.hip && ./valu_ceiling > profiles/valu_ceiling.json
//
// Method: every block is pinned one (or two) per CU with a large dynamic LDS request, block size = 256 x k
// lanes = k waves on each of the CU's 4 SIMDs; every wave runs ITERS x 64 instances of the instruction in
// straight-line inline asm.  Rate = waves x instructions / event time, reported as G wave-instructions/s for
// the whole chip and as cycles per instruction per SIMD at the clock measured with s_memrealtime (100 MHz)
// against the shader clock counter (s_memtime).
#include <hip/hip_runtime.h>
#include <cstring>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 register operands a..h (32-bit), two 64-bit pairs p, q; x, y loop-invariant sources
// 64 instances per loop iteration in ONE asm statement (the compiler pads every asm statement with an s_nop)
#define R8(S) S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n"
#define G8(S0, S1, S2, S3, S4, S5, S6, S7) S0 "\n" S1 "\n" S2 "\n" S3 "\n" S4 "\n" S5 "\n" S6 "\n" S7 "\n"

enum Op {
    OP_BITOP3, OP_AND, OP_ADD, OP_ADDCO_PAIR, OP_LSHL_OR, OP_ALIGNBIT, OP_ADD3, OP_OR3, OP_MAD24, OP_BCNT, OP_LSHL, OP_LSHL64,
    OP_CNDMASK, OP_BFREV, OP_MOV, OP_ADD_F64, OP_FMA_F64, OP_DS_READ_B32, OP_DS_READ_B64, OP_LSHL_ADD_U64, OP_XAD, OP_CMP_CND, OP_BFE, OP_AND_SDWA, OP_ADD_SDWA, OP_ADDCO, OP_ADDC, OP_FFBH, OP_CMP, OP_AND_OR, OP_LSHR, OP_NOT, OP_SUB, OP_MIN, OP_OR, OP_XNOR, OP_BFI, OP_PERM, OP_LSHLADD, OP_PK_ADD16, OP_PK_SUB16, OP_PK_LSHL16, OP_PK_LSHR16, OP_LSHL_SDWA, OP_ASHR, OP_AND_LIT, OP_XOR, OP_SUBREV, OP_MYERS2, OP_COUNT
};
static const char* const kOpName[OP_COUNT] = {
    "v_bitop3_b32", "v_and_b32", "v_add_u32", "v_add_co_u32+v_addc_co_u32", "v_lshl_or_b32", "v_alignbit_b32", "v_add3_u32", "v_or3_b32",
    "v_mad_u32_u24", "v_bcnt_u32_b32", "v_lshlrev_b32", "v_lshlrev_b64", "v_cndmask_b32", "v_bfrev_b32", "v_mov_b32", "v_add_f64",
    "v_fma_f64", "ds_read_b32", "ds_read_b64", "v_lshl_add_u64", "v_xad_u32", "v_cmp_ne_u32+v_cndmask_b32", "v_bfe_u32", "v_and_b32_sdwa(byte_sel)", "v_add_u32_sdwa(word_sel)", "v_add_co_u32(alone)", "v_addc_co_u32(alone)", "v_ffbh_u32", "v_cmp_lt_i32(vcc)", "v_and_or_b32", "v_lshrrev_b32", "v_not_b32", "v_sub_u32", "v_min_i32", "v_or_b32", "v_xnor_b32", "v_bfi_b32", "v_perm_b32", "v_lshl_add_u32", "v_pk_add_u16", "v_pk_sub_u16", "v_pk_lshlrev_b16", "v_pk_lshrrev_b16", "v_lshlrev_b32_sdwa(byte_sel)", "v_ashrrev_i32", "v_and_b32(32-bit literal)", "v_xor_b32", "v_subrev_u32", "myers_step<2> + move_bits (C++, 27 VALU)"};
// instructions per asm instance (the add/addc pair counts two)
static const int kOpInstr[OP_COUNT] = {1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

#define A1(INS) asm volatile(INS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(x), "v"(y) : "vcc")

template <int OP, bool DEP>
__global__ void k_issue(uint32_t* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
    extern __shared__ uint32_t lds[];
    uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    const uint32_t x = out[0], y = out[1];
    unsigned long long p = a, q = b, r = c, s = d;
    if (OP == OP_DS_READ_B32 || OP == OP_DS_READ_B64) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = ((i * 8u) & 0x3FF8u);  // every word holds a valid byte address
        __syncthreads();
        a = (threadIdx.x * 8u) & 0x3FF8u; b = a; c = a; d = a; e = a; f = a; g = a; h = a;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#define I_DEP(INS_A) A1(R8(R8(INS_A)));
#define I_IND(IA, IB, IC, ID, IE, IF, IG, IH) A1(R8(G8(IA, IB, IC, ID, IE, IF, IG, IH)));
        if constexpr (OP == OP_BITOP3) {
            if constexpr (DEP) { I_DEP("v_bitop3_b32 %0, %0, %12, %13 bitop3:0x96") }
            else { I_IND("v_bitop3_b32 %0, %0, %12, %13 bitop3:0x96", "v_bitop3_b32 %1, %1, %12, %13 bitop3:0x96", "v_bitop3_b32 %2, %2, %12, %13 bitop3:0x96",
                         "v_bitop3_b32 %3, %3, %12, %13 bitop3:0x96", "v_bitop3_b32 %4, %4, %12, %13 bitop3:0x96", "v_bitop3_b32 %5, %5, %12, %13 bitop3:0x96",
                         "v_bitop3_b32 %6, %6, %12, %13 bitop3:0x96", "v_bitop3_b32 %7, %7, %12, %13 bitop3:0x96") }
        } else if constexpr (OP == OP_AND) {
            if constexpr (DEP) { I_DEP("v_xor_b32 %0, %0, %12") }
            else { I_IND("v_xor_b32 %0, %0, %12", "v_xor_b32 %1, %1, %12", "v_xor_b32 %2, %2, %12", "v_xor_b32 %3, %3, %12", "v_xor_b32 %4, %4, %12",
                         "v_xor_b32 %5, %5, %12", "v_xor_b32 %6, %6, %12", "v_xor_b32 %7, %7, %12") }
        } else if constexpr (OP == OP_ADD) {
            if constexpr (DEP) { I_DEP("v_add_u32 %0, %0, %12") }
            else { I_IND("v_add_u32 %0, %0, %12", "v_add_u32 %1, %1, %12", "v_add_u32 %2, %2, %12", "v_add_u32 %3, %3, %12", "v_add_u32 %4, %4, %12",
                         "v_add_u32 %5, %5, %12", "v_add_u32 %6, %6, %12", "v_add_u32 %7, %7, %12") }
        } else if constexpr (OP == OP_ADDCO_PAIR) {
            if constexpr (DEP) { I_DEP("v_add_co_u32 %0, vcc, %0, %12\n v_addc_co_u32 %1, vcc, %1, %13, vcc") }
            else { I_IND("v_add_co_u32 %0, vcc, %0, %12\n v_addc_co_u32 %1, vcc, %1, %13, vcc", "v_add_co_u32 %2, vcc, %2, %12\n v_addc_co_u32 %3, vcc, %3, %13, vcc",
                         "v_add_co_u32 %4, vcc, %4, %12\n v_addc_co_u32 %5, vcc, %5, %13, vcc", "v_add_co_u32 %6, vcc, %6, %12\n v_addc_co_u32 %7, vcc, %7, %13, vcc",
                         "v_add_co_u32 %0, vcc, %0, %12\n v_addc_co_u32 %1, vcc, %1, %13, vcc", "v_add_co_u32 %2, vcc, %2, %12\n v_addc_co_u32 %3, vcc, %3, %13, vcc",
                         "v_add_co_u32 %4, vcc, %4, %12\n v_addc_co_u32 %5, vcc, %5, %13, vcc", "v_add_co_u32 %6, vcc, %6, %12\n v_addc_co_u32 %7, vcc, %7, %13, vcc") }
        } else if constexpr (OP == OP_LSHL_OR) {
            if constexpr (DEP) { I_DEP("v_lshl_or_b32 %0, %0, 1, %12") }
            else { I_IND("v_lshl_or_b32 %0, %0, 1, %12", "v_lshl_or_b32 %1, %1, 1, %12", "v_lshl_or_b32 %2, %2, 1, %12", "v_lshl_or_b32 %3, %3, 1, %12",
                         "v_lshl_or_b32 %4, %4, 1, %12", "v_lshl_or_b32 %5, %5, 1, %12", "v_lshl_or_b32 %6, %6, 1, %12", "v_lshl_or_b32 %7, %7, 1, %12") }
        } else if constexpr (OP == OP_ALIGNBIT) {
            if constexpr (DEP) { I_DEP("v_alignbit_b32 %0, %0, %12, 31") }
            else { I_IND("v_alignbit_b32 %0, %0, %12, 31", "v_alignbit_b32 %1, %1, %12, 31", "v_alignbit_b32 %2, %2, %12, 31", "v_alignbit_b32 %3, %3, %12, 31",
                         "v_alignbit_b32 %4, %4, %12, 31", "v_alignbit_b32 %5, %5, %12, 31", "v_alignbit_b32 %6, %6, %12, 31", "v_alignbit_b32 %7, %7, %12, 31") }
        } else if constexpr (OP == OP_ADD3) {
            if constexpr (DEP) { I_DEP("v_add3_u32 %0, %0, %12, %13") }
            else { I_IND("v_add3_u32 %0, %0, %12, %13", "v_add3_u32 %1, %1, %12, %13", "v_add3_u32 %2, %2, %12, %13", "v_add3_u32 %3, %3, %12, %13",
                         "v_add3_u32 %4, %4, %12, %13", "v_add3_u32 %5, %5, %12, %13", "v_add3_u32 %6, %6, %12, %13", "v_add3_u32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_OR3) {
            if constexpr (DEP) { I_DEP("v_or3_b32 %0, %0, %12, %13") }
            else { I_IND("v_or3_b32 %0, %0, %12, %13", "v_or3_b32 %1, %1, %12, %13", "v_or3_b32 %2, %2, %12, %13", "v_or3_b32 %3, %3, %12, %13",
                         "v_or3_b32 %4, %4, %12, %13", "v_or3_b32 %5, %5, %12, %13", "v_or3_b32 %6, %6, %12, %13", "v_or3_b32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_MAD24) {
            if constexpr (DEP) { I_DEP("v_mad_u32_u24 %0, %0, %12, %13") }
            else { I_IND("v_mad_u32_u24 %0, %0, %12, %13", "v_mad_u32_u24 %1, %1, %12, %13", "v_mad_u32_u24 %2, %2, %12, %13", "v_mad_u32_u24 %3, %3, %12, %13",
                         "v_mad_u32_u24 %4, %4, %12, %13", "v_mad_u32_u24 %5, %5, %12, %13", "v_mad_u32_u24 %6, %6, %12, %13", "v_mad_u32_u24 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_BCNT) {
            if constexpr (DEP) { I_DEP("v_bcnt_u32_b32 %0, %0, %12") }
            else { I_IND("v_bcnt_u32_b32 %0, %0, %12", "v_bcnt_u32_b32 %1, %1, %12", "v_bcnt_u32_b32 %2, %2, %12", "v_bcnt_u32_b32 %3, %3, %12",
                         "v_bcnt_u32_b32 %4, %4, %12", "v_bcnt_u32_b32 %5, %5, %12", "v_bcnt_u32_b32 %6, %6, %12", "v_bcnt_u32_b32 %7, %7, %12") }
        } else if constexpr (OP == OP_LSHL) {
            if constexpr (DEP) { I_DEP("v_lshlrev_b32 %0, 1, %0") }
            else { I_IND("v_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %1, 1, %1", "v_lshlrev_b32 %2, 1, %2", "v_lshlrev_b32 %3, 1, %3", "v_lshlrev_b32 %4, 1, %4",
                         "v_lshlrev_b32 %5, 1, %5", "v_lshlrev_b32 %6, 1, %6", "v_lshlrev_b32 %7, 1, %7") }
        } else if constexpr (OP == OP_LSHL64) {
            if constexpr (DEP) { I_DEP("v_lshlrev_b64 %8, 1, %8") }
            else { I_IND("v_lshlrev_b64 %8, 1, %8", "v_lshlrev_b64 %9, 1, %9", "v_lshlrev_b64 %10, 1, %10", "v_lshlrev_b64 %11, 1, %11", "v_lshlrev_b64 %8, 1, %8",
                         "v_lshlrev_b64 %9, 1, %9", "v_lshlrev_b64 %10, 1, %10", "v_lshlrev_b64 %11, 1, %11") }
        } else if constexpr (OP == OP_CNDMASK) {
            if constexpr (DEP) { I_DEP("v_cndmask_b32 %0, %0, %12, vcc") }
            else { I_IND("v_cndmask_b32 %0, %0, %12, vcc", "v_cndmask_b32 %1, %1, %12, vcc", "v_cndmask_b32 %2, %2, %12, vcc", "v_cndmask_b32 %3, %3, %12, vcc",
                         "v_cndmask_b32 %4, %4, %12, vcc", "v_cndmask_b32 %5, %5, %12, vcc", "v_cndmask_b32 %6, %6, %12, vcc", "v_cndmask_b32 %7, %7, %12, vcc") }
        } else if constexpr (OP == OP_BFREV) {
            if constexpr (DEP) { I_DEP("v_bfrev_b32 %0, %0") }
            else { I_IND("v_bfrev_b32 %0, %0", "v_bfrev_b32 %1, %1", "v_bfrev_b32 %2, %2", "v_bfrev_b32 %3, %3", "v_bfrev_b32 %4, %4", "v_bfrev_b32 %5, %5",
                         "v_bfrev_b32 %6, %6", "v_bfrev_b32 %7, %7") }
        } else if constexpr (OP == OP_MOV) {
            if constexpr (DEP) { I_DEP("v_mov_b32 %0, %0") }
            else { I_IND("v_mov_b32 %0, %12", "v_mov_b32 %1, %12", "v_mov_b32 %2, %12", "v_mov_b32 %3, %12", "v_mov_b32 %4, %12", "v_mov_b32 %5, %12",
                         "v_mov_b32 %6, %12", "v_mov_b32 %7, %12") }
        } else if constexpr (OP == OP_ADD_F64) {
            if constexpr (DEP) { I_DEP("v_add_f64 %8, %8, %8") }
            else { I_IND("v_add_f64 %8, %8, %8", "v_add_f64 %9, %9, %9", "v_add_f64 %10, %10, %10", "v_add_f64 %11, %11, %11", "v_add_f64 %8, %8, %8",
                         "v_add_f64 %9, %9, %9", "v_add_f64 %10, %10, %10", "v_add_f64 %11, %11, %11") }
        } else if constexpr (OP == OP_FMA_F64) {
            if constexpr (DEP) { I_DEP("v_fma_f64 %8, %8, %8, %8") }
            else { I_IND("v_fma_f64 %8, %8, %8, %8", "v_fma_f64 %9, %9, %9, %9", "v_fma_f64 %10, %10, %10, %10", "v_fma_f64 %11, %11, %11, %11",
                         "v_fma_f64 %8, %8, %8, %8", "v_fma_f64 %9, %9, %9, %9", "v_fma_f64 %10, %10, %10, %10", "v_fma_f64 %11, %11, %11, %11") }
        } else if constexpr (OP == OP_LSHL_ADD_U64) {
            if constexpr (DEP) { I_DEP("v_lshl_add_u64 %8, %8, 0, %9") }
            else { I_IND("v_lshl_add_u64 %8, %8, 0, %8", "v_lshl_add_u64 %9, %9, 0, %9", "v_lshl_add_u64 %10, %10, 0, %10", "v_lshl_add_u64 %11, %11, 0, %11",
                         "v_lshl_add_u64 %8, %8, 0, %8", "v_lshl_add_u64 %9, %9, 0, %9", "v_lshl_add_u64 %10, %10, 0, %10", "v_lshl_add_u64 %11, %11, 0, %11") }
        } else if constexpr (OP == OP_XAD) {
            if constexpr (DEP) { I_DEP("v_xad_u32 %0, %0, %12, %13") }
            else { I_IND("v_xad_u32 %0, %0, %12, %13", "v_xad_u32 %1, %1, %12, %13", "v_xad_u32 %2, %2, %12, %13", "v_xad_u32 %3, %3, %12, %13",
                         "v_xad_u32 %4, %4, %12, %13", "v_xad_u32 %5, %5, %12, %13", "v_xad_u32 %6, %6, %12, %13", "v_xad_u32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_CMP_CND) {
            if constexpr (DEP) { I_DEP("v_cmp_ne_u32 vcc, %0, %12\n v_cndmask_b32 %0, %0, %13, vcc") }
            else { I_IND("v_cmp_ne_u32 vcc, %0, %12\n v_cndmask_b32 %0, %0, %13, vcc", "v_cmp_ne_u32 vcc, %1, %12\n v_cndmask_b32 %1, %1, %13, vcc",
                         "v_cmp_ne_u32 vcc, %2, %12\n v_cndmask_b32 %2, %2, %13, vcc", "v_cmp_ne_u32 vcc, %3, %12\n v_cndmask_b32 %3, %3, %13, vcc",
                         "v_cmp_ne_u32 vcc, %4, %12\n v_cndmask_b32 %4, %4, %13, vcc", "v_cmp_ne_u32 vcc, %5, %12\n v_cndmask_b32 %5, %5, %13, vcc",
                         "v_cmp_ne_u32 vcc, %6, %12\n v_cndmask_b32 %6, %6, %13, vcc", "v_cmp_ne_u32 vcc, %7, %12\n v_cndmask_b32 %7, %7, %13, vcc") }
        } else if constexpr (OP == OP_BFE) {
            if constexpr (DEP) { I_DEP("v_bfe_u32 %0, %0, 4, 28") }
            else { I_IND("v_bfe_u32 %0, %0, 4, 28", "v_bfe_u32 %1, %1, 4, 28", "v_bfe_u32 %2, %2, 4, 28", "v_bfe_u32 %3, %3, 4, 28", "v_bfe_u32 %4, %4, 4, 28", "v_bfe_u32 %5, %5, 4, 28", "v_bfe_u32 %6, %6, 4, 28", "v_bfe_u32 %7, %7, 4, 28") }
        } else if constexpr (OP == OP_AND_SDWA) {
            if constexpr (DEP) { I_DEP("v_and_b32_sdwa %0, %0, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") }
            else { I_IND("v_and_b32_sdwa %0, %0, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %1, %1, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %2, %2, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %3, %3, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %4, %4, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %5, %5, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %6, %6, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "v_and_b32_sdwa %7, %7, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") }
        } else if constexpr (OP == OP_ADD_SDWA) {
            if constexpr (DEP) { I_DEP("v_add_u32_sdwa %0, %0, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") }
            else { I_IND("v_add_u32_sdwa %0, %0, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %1, %1, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %2, %2, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %3, %3, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %4, %4, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %5, %5, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %6, %6, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "v_add_u32_sdwa %7, %7, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") }
        } else if constexpr (OP == OP_ADDCO) {
            if constexpr (DEP) { I_DEP("v_add_co_u32 %0, vcc, %0, %12") }
            else { I_IND("v_add_co_u32 %0, vcc, %0, %12", "v_add_co_u32 %1, vcc, %1, %12", "v_add_co_u32 %2, vcc, %2, %12", "v_add_co_u32 %3, vcc, %3, %12", "v_add_co_u32 %4, vcc, %4, %12", "v_add_co_u32 %5, vcc, %5, %12", "v_add_co_u32 %6, vcc, %6, %12", "v_add_co_u32 %7, vcc, %7, %12") }
        } else if constexpr (OP == OP_ADDC) {
            if constexpr (DEP) { I_DEP("v_addc_co_u32 %0, vcc, %0, %12, vcc") }
            else { I_IND("v_addc_co_u32 %0, vcc, %0, %12, vcc", "v_addc_co_u32 %1, vcc, %1, %12, vcc", "v_addc_co_u32 %2, vcc, %2, %12, vcc", "v_addc_co_u32 %3, vcc, %3, %12, vcc", "v_addc_co_u32 %4, vcc, %4, %12, vcc", "v_addc_co_u32 %5, vcc, %5, %12, vcc", "v_addc_co_u32 %6, vcc, %6, %12, vcc", "v_addc_co_u32 %7, vcc, %7, %12, vcc") }
        } else if constexpr (OP == OP_FFBH) {
            if constexpr (DEP) { I_DEP("v_ffbh_u32 %0, %0") }
            else { I_IND("v_ffbh_u32 %0, %0", "v_ffbh_u32 %1, %1", "v_ffbh_u32 %2, %2", "v_ffbh_u32 %3, %3", "v_ffbh_u32 %4, %4", "v_ffbh_u32 %5, %5", "v_ffbh_u32 %6, %6", "v_ffbh_u32 %7, %7") }
        } else if constexpr (OP == OP_CMP) {
            if constexpr (DEP) { I_DEP("v_cmp_lt_i32 vcc, %0, %12") }
            else { I_IND("v_cmp_lt_i32 vcc, %0, %12", "v_cmp_lt_i32 vcc, %1, %12", "v_cmp_lt_i32 vcc, %2, %12", "v_cmp_lt_i32 vcc, %3, %12", "v_cmp_lt_i32 vcc, %4, %12", "v_cmp_lt_i32 vcc, %5, %12", "v_cmp_lt_i32 vcc, %6, %12", "v_cmp_lt_i32 vcc, %7, %12") }
        } else if constexpr (OP == OP_AND_OR) {
            if constexpr (DEP) { I_DEP("v_and_or_b32 %0, %0, %12, %13") }
            else { I_IND("v_and_or_b32 %0, %0, %12, %13", "v_and_or_b32 %1, %1, %12, %13", "v_and_or_b32 %2, %2, %12, %13", "v_and_or_b32 %3, %3, %12, %13", "v_and_or_b32 %4, %4, %12, %13", "v_and_or_b32 %5, %5, %12, %13", "v_and_or_b32 %6, %6, %12, %13", "v_and_or_b32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_LSHR) {
            if constexpr (DEP) { I_DEP("v_lshrrev_b32 %0, 1, %0") }
            else { I_IND("v_lshrrev_b32 %0, 1, %0", "v_lshrrev_b32 %1, 1, %1", "v_lshrrev_b32 %2, 1, %2", "v_lshrrev_b32 %3, 1, %3", "v_lshrrev_b32 %4, 1, %4", "v_lshrrev_b32 %5, 1, %5", "v_lshrrev_b32 %6, 1, %6", "v_lshrrev_b32 %7, 1, %7") }
        } else if constexpr (OP == OP_NOT) {
            if constexpr (DEP) { I_DEP("v_not_b32 %0, %0") }
            else { I_IND("v_not_b32 %0, %0", "v_not_b32 %1, %1", "v_not_b32 %2, %2", "v_not_b32 %3, %3", "v_not_b32 %4, %4", "v_not_b32 %5, %5", "v_not_b32 %6, %6", "v_not_b32 %7, %7") }
        } else if constexpr (OP == OP_SUB) {
            if constexpr (DEP) { I_DEP("v_sub_u32 %0, %0, %12") }
            else { I_IND("v_sub_u32 %0, %0, %12", "v_sub_u32 %1, %1, %12", "v_sub_u32 %2, %2, %12", "v_sub_u32 %3, %3, %12", "v_sub_u32 %4, %4, %12", "v_sub_u32 %5, %5, %12", "v_sub_u32 %6, %6, %12", "v_sub_u32 %7, %7, %12") }
        } else if constexpr (OP == OP_MIN) {
            if constexpr (DEP) { I_DEP("v_min_i32 %0, %0, %12") }
            else { I_IND("v_min_i32 %0, %0, %12", "v_min_i32 %1, %1, %12", "v_min_i32 %2, %2, %12", "v_min_i32 %3, %3, %12", "v_min_i32 %4, %4, %12", "v_min_i32 %5, %5, %12", "v_min_i32 %6, %6, %12", "v_min_i32 %7, %7, %12") }
        } else if constexpr (OP == OP_OR) {
            if constexpr (DEP) { I_DEP("v_or_b32 %0, %0, %12") }
            else { I_IND("v_or_b32 %0, %0, %12", "v_or_b32 %1, %1, %12", "v_or_b32 %2, %2, %12", "v_or_b32 %3, %3, %12", "v_or_b32 %4, %4, %12", "v_or_b32 %5, %5, %12", "v_or_b32 %6, %6, %12", "v_or_b32 %7, %7, %12") }
        } else if constexpr (OP == OP_XNOR) {
            if constexpr (DEP) { I_DEP("v_xnor_b32 %0, %0, %12") }
            else { I_IND("v_xnor_b32 %0, %0, %12", "v_xnor_b32 %1, %1, %12", "v_xnor_b32 %2, %2, %12", "v_xnor_b32 %3, %3, %12", "v_xnor_b32 %4, %4, %12", "v_xnor_b32 %5, %5, %12", "v_xnor_b32 %6, %6, %12", "v_xnor_b32 %7, %7, %12") }
        } else if constexpr (OP == OP_BFI) {
            if constexpr (DEP) { I_DEP("v_bfi_b32 %0, %0, %12, %13") }
            else { I_IND("v_bfi_b32 %0, %0, %12, %13", "v_bfi_b32 %1, %1, %12, %13", "v_bfi_b32 %2, %2, %12, %13", "v_bfi_b32 %3, %3, %12, %13", "v_bfi_b32 %4, %4, %12, %13", "v_bfi_b32 %5, %5, %12, %13", "v_bfi_b32 %6, %6, %12, %13", "v_bfi_b32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_PERM) {
            if constexpr (DEP) { I_DEP("v_perm_b32 %0, %0, %12, %13") }
            else { I_IND("v_perm_b32 %0, %0, %12, %13", "v_perm_b32 %1, %1, %12, %13", "v_perm_b32 %2, %2, %12, %13", "v_perm_b32 %3, %3, %12, %13", "v_perm_b32 %4, %4, %12, %13", "v_perm_b32 %5, %5, %12, %13", "v_perm_b32 %6, %6, %12, %13", "v_perm_b32 %7, %7, %12, %13") }
        } else if constexpr (OP == OP_LSHLADD) {
            if constexpr (DEP) { I_DEP("v_lshl_add_u32 %0, %0, 1, %12") }
            else { I_IND("v_lshl_add_u32 %0, %0, 1, %12", "v_lshl_add_u32 %1, %1, 1, %12", "v_lshl_add_u32 %2, %2, 1, %12", "v_lshl_add_u32 %3, %3, 1, %12", "v_lshl_add_u32 %4, %4, 1, %12", "v_lshl_add_u32 %5, %5, 1, %12", "v_lshl_add_u32 %6, %6, 1, %12", "v_lshl_add_u32 %7, %7, 1, %12") }
        } else if constexpr (OP == OP_PK_ADD16) {
            if constexpr (DEP) { I_DEP("v_pk_add_u16 %0, %0, %12") }
            else { I_IND("v_pk_add_u16 %0, %0, %12", "v_pk_add_u16 %1, %1, %12", "v_pk_add_u16 %2, %2, %12", "v_pk_add_u16 %3, %3, %12", "v_pk_add_u16 %4, %4, %12", "v_pk_add_u16 %5, %5, %12", "v_pk_add_u16 %6, %6, %12", "v_pk_add_u16 %7, %7, %12") }
        } else if constexpr (OP == OP_PK_SUB16) {
            if constexpr (DEP) { I_DEP("v_pk_sub_u16 %0, %0, %12") }
            else { I_IND("v_pk_sub_u16 %0, %0, %12", "v_pk_sub_u16 %1, %1, %12", "v_pk_sub_u16 %2, %2, %12", "v_pk_sub_u16 %3, %3, %12", "v_pk_sub_u16 %4, %4, %12", "v_pk_sub_u16 %5, %5, %12", "v_pk_sub_u16 %6, %6, %12", "v_pk_sub_u16 %7, %7, %12") }
        } else if constexpr (OP == OP_PK_LSHL16) {
            if constexpr (DEP) { I_DEP("v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]") }
            else { I_IND("v_pk_lshlrev_b16 %0, 1, %0 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %1, 1, %1 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %2, 1, %2 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %3, 1, %3 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %4, 1, %4 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %5, 1, %5 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %6, 1, %6 op_sel_hi:[0,1]", "v_pk_lshlrev_b16 %7, 1, %7 op_sel_hi:[0,1]") }
        } else if constexpr (OP == OP_ASHR) {
            if constexpr (DEP) { I_DEP("v_ashrrev_i32 %0, 1, %0") }
            else { I_IND("v_ashrrev_i32 %0, 1, %0", "v_ashrrev_i32 %1, 1, %1", "v_ashrrev_i32 %2, 1, %2", "v_ashrrev_i32 %3, 1, %3", "v_ashrrev_i32 %4, 1, %4", "v_ashrrev_i32 %5, 1, %5", "v_ashrrev_i32 %6, 1, %6", "v_ashrrev_i32 %7, 1, %7") }
        } else if constexpr (OP == OP_AND_LIT) {
            if constexpr (DEP) { I_DEP("v_and_b32 %0, 0x40004000, %0") }
            else { I_IND("v_and_b32 %0, 0x40004000, %0", "v_and_b32 %1, 0x40004000, %1", "v_and_b32 %2, 0x40004000, %2", "v_and_b32 %3, 0x40004000, %3", "v_and_b32 %4, 0x40004000, %4", "v_and_b32 %5, 0x40004000, %5", "v_and_b32 %6, 0x40004000, %6", "v_and_b32 %7, 0x40004000, %7") }
        } else if constexpr (OP == OP_XOR) {
            if constexpr (DEP) { I_DEP("v_xor_b32 %0, %0, %12") }
            else { I_IND("v_xor_b32 %0, %0, %12", "v_xor_b32 %1, %1, %12", "v_xor_b32 %2, %2, %12", "v_xor_b32 %3, %3, %12", "v_xor_b32 %4, %4, %12", "v_xor_b32 %5, %5, %12", "v_xor_b32 %6, %6, %12", "v_xor_b32 %7, %7, %12") }
        } else if constexpr (OP == OP_SUBREV) {
            if constexpr (DEP) { I_DEP("v_subrev_u32 %0, %12, %0") }
            else { I_IND("v_subrev_u32 %0, %12, %0", "v_subrev_u32 %1, %12, %1", "v_subrev_u32 %2, %12, %2", "v_subrev_u32 %3, %12, %3", "v_subrev_u32 %4, %12, %4", "v_subrev_u32 %5, %12, %5", "v_subrev_u32 %6, %12, %6", "v_subrev_u32 %7, %12, %7") }
        } else if constexpr (OP == OP_PK_LSHR16) {
            if constexpr (DEP) { I_DEP("v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]") }
            else { I_IND("v_pk_lshrrev_b16 %0, 15, %0 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %1, 15, %1 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %2, 15, %2 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %3, 15, %3 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %4, 15, %4 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %5, 15, %5 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %6, 15, %6 op_sel_hi:[0,1]", "v_pk_lshrrev_b16 %7, 15, %7 op_sel_hi:[0,1]") }
        } else if constexpr (OP == OP_LSHL_SDWA) {
            if constexpr (DEP) { I_DEP("v_lshlrev_b32_sdwa %0, %13, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") }
            else { I_IND("v_lshlrev_b32_sdwa %0, %13, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %1, %13, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %2, %13, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %3, %13, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %4, %13, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %5, %13, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %6, %13, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", "v_lshlrev_b32_sdwa %7, %13, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") }
        } else if constexpr (OP == OP_DS_READ_B32) {
            if constexpr (DEP) { I_DEP("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)") }
            else { A1(R8("ds_read_b32 %0, %0\n ds_read_b32 %1, %1\n ds_read_b32 %2, %2\n ds_read_b32 %3, %3\n ds_read_b32 %4, %4\n ds_read_b32 %5, %5\n"
                         "ds_read_b32 %6, %6\n ds_read_b32 %7, %7\n s_waitcnt lgkmcnt(0)")); }
        } else if constexpr (OP == OP_DS_READ_B64) {
            if constexpr (DEP) { I_DEP("ds_read_b64 %8, %0\n s_waitcnt lgkmcnt(0)") }
            else { A1(R8("ds_read_b64 %8, %0\n ds_read_b64 %9, %1\n ds_read_b64 %10, %2\n ds_read_b64 %11, %3\n ds_read_b64 %8, %4\n ds_read_b64 %9, %5\n"
                         "ds_read_b64 %10, %6\n ds_read_b64 %11, %7\n s_waitcnt lgkmcnt(0)")); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)p ^ (uint32_t)q ^ (uint32_t)r ^ (uint32_t)s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// the annotate kernels' own inner step in C++ (two-word Myers column + move bits, as in bb_kernels.h), DEP = one
// column chain per lane (as in the kernels), !DEP = two independent chains per lane interleaved
template <int TT> __device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }
struct col2 { uint32_t pv0, pv1, mv0, mv1, acc; };
__device__ __forceinline__ void myers2(col2& s, uint32_t eq0, uint32_t eq1) {
    const uint32_t x0 = eq0 & s.pv0, x1 = eq1 & s.pv1;
    const unsigned long long s0 = (unsigned long long)x0 + s.pv0;
    const uint32_t s1 = x1 + s.pv1 + (uint32_t)(s0 >> 32);
    const uint32_t d00 = bitop3<0xBE>((uint32_t)s0, s.pv0, eq0) | s.mv0, d01 = bitop3<0xBE>(s1, s.pv1, eq1) | s.mv1;
    const uint32_t ph0 = bitop3<0xF1>(s.mv0, d00, s.pv0), ph1 = bitop3<0xF1>(s.mv1, d01, s.pv1);
    const uint32_t mh0 = s.pv0 & d00, mh1 = s.pv1 & d01;
    const uint32_t phs1 = (ph1 << 1) | (ph0 >> 31), phs0 = ph0 << 1, mhs1 = (mh1 << 1) | (mh0 >> 31), mhs0 = mh0 << 1;
    s.pv0 = bitop3<0xF1>(mhs0, d00, phs0); s.pv1 = bitop3<0xF1>(mhs1, d01, phs1);
    s.mv0 = phs0 & d00; s.mv1 = phs1 & d01;
    s.acc += bitop3<0x15>(d00, eq0, ph0) ^ bitop3<0x3A>(d00, eq0, ph0) ^ bitop3<0x15>(d01, eq1, ph1) ^ bitop3<0x3A>(d01, eq1, ph1);
}
struct col64 { unsigned long long pv, mv; uint32_t acc; };
__device__ __forceinline__ void myers64(col64& s, unsigned long long eq) {
    const unsigned long long x = eq & s.pv;
    const unsigned long long d0 = (((x + s.pv) ^ s.pv) | eq) | s.mv;
    const unsigned long long ph = s.mv | ~(d0 | s.pv), mh = s.pv & d0;
    const unsigned long long phs = ph << 1, mhs = mh << 1;
    s.pv = mhs | ~(d0 | phs);
    s.mv = phs & d0;
    const unsigned long long isM = d0 & eq, l = ~(isM | ph), hh = (ph & ~isM) | (l & d0);
    s.acc += (uint32_t)l ^ (uint32_t)(l >> 32) ^ (uint32_t)hh ^ (uint32_t)(hh >> 32);
}
// hybrid: boolean steps as 32-bit v_bitop3 on the halves, the addition and the two shifts as 64-bit instructions
__device__ __forceinline__ void myers2h(col2& s, uint32_t eq0, uint32_t eq1) {
    const uint32_t x0 = eq0 & s.pv0, x1 = eq1 & s.pv1;
    const unsigned long long sum = (((unsigned long long)x1 << 32) | x0) + (((unsigned long long)s.pv1 << 32) | s.pv0);
    const uint32_t d00 = bitop3<0xBE>((uint32_t)sum, s.pv0, eq0) | s.mv0, d01 = bitop3<0xBE>((uint32_t)(sum >> 32), s.pv1, eq1) | s.mv1;
    const uint32_t ph0 = bitop3<0xF1>(s.mv0, d00, s.pv0), ph1 = bitop3<0xF1>(s.mv1, d01, s.pv1);
    const uint32_t mh0 = s.pv0 & d00, mh1 = s.pv1 & d01;
    unsigned long long phs, mhs;
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(phs) : "v"(((unsigned long long)ph1 << 32) | ph0));
    asm("v_lshlrev_b64 %0, 1, %1" : "=v"(mhs) : "v"(((unsigned long long)mh1 << 32) | mh0));
    const uint32_t phs0 = (uint32_t)phs, phs1 = (uint32_t)(phs >> 32), mhs0 = (uint32_t)mhs, mhs1 = (uint32_t)(mhs >> 32);
    s.pv0 = bitop3<0xF1>(mhs0, d00, phs0); s.pv1 = bitop3<0xF1>(mhs1, d01, phs1);
    s.mv0 = phs0 & d00; s.mv1 = phs1 & d01;
    s.acc += bitop3<0x15>(d00, eq0, ph0) ^ bitop3<0x3A>(d00, eq0, ph0) ^ bitop3<0x15>(d01, eq1, ph1) ^ bitop3<0x3A>(d01, eq1, ph1);
}
__global__ void k_myers2h(uint32_t* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
    col2 A = {threadIdx.x * 2654435761u, ~threadIdx.x, 0u, 0u, 0u};
    uint32_t e0 = out[0] ^ threadIdx.x, e1 = out[1] + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            myers2h(A, e0, e1);
            e0 = (e0 >> 1) | (e0 << 31); e1 ^= e0;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = A.pv0 ^ A.pv1 ^ A.mv0 ^ A.mv1 ^ A.acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
__global__ void k_myers64(uint32_t* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
    col64 A = {((unsigned long long)threadIdx.x * 2654435761u) | ((unsigned long long)~threadIdx.x << 32), 0ull, 0u};
    uint32_t e0 = out[0] ^ threadIdx.x, e1 = out[1] + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            myers64(A, ((unsigned long long)e1 << 32) | e0);
            e0 = (e0 >> 1) | (e0 << 31); e1 ^= e0;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)A.pv ^ (uint32_t)(A.pv >> 32) ^ (uint32_t)A.mv ^ (uint32_t)(A.mv >> 32) ^ A.acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <bool DEP>
__global__ void k_myers(uint32_t* __restrict__ out, int iters, unsigned long long* __restrict__ clk) {
    col2 A = {threadIdx.x * 2654435761u, ~threadIdx.x, 0u, 0u, 0u}, B = {threadIdx.x * 40503u, threadIdx.x, 0u, 0u, 0u};
    uint32_t e0 = out[0] ^ threadIdx.x, e1 = out[1] + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            myers2(A, e0, e1);
            if (!DEP) myers2(B, e1, e0);
            e0 = (e0 >> 1) | (e0 << 31); e1 ^= e0;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = A.pv0 ^ A.pv1 ^ A.mv0 ^ A.mv1 ^ A.acc ^ B.pv0 ^ B.pv1 ^ B.mv0 ^ B.mv1 ^ B.acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

struct Result { double ginstr_s, cyc_per_instr_simd, mhz; };

template <typename K>
static Result run(K kern, int wps, int instr_per_iter, int iters, int n_cus, uint32_t* d_out, unsigned long long* d_clk) {
    // wps waves per SIMD: 1..4 -> one block of 256*wps lanes per CU; 6, 8 -> two blocks of 128*wps lanes per CU
    const int per_cu = wps <= 4 ? 1 : 2;
    const int threads = 256 * wps / per_cu;
    const size_t lds = per_cu == 1 ? 100 * 1024 : 70 * 1024;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = n_cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, 16, d_clk);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    unsigned long long clk[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_out, iters, d_clk);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CHECK(hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost)); }
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    const double waves = (double)blocks * threads / 64.0;
    const double instr = (double)instr_per_iter * iters;
    Result r;
    r.ginstr_s = waves * instr / (best * 1e-3) / 1e9;
    // shader cycles of block 0 / wall time of block 0 (wall_clock64 ticks at 100 MHz)
    r.mhz = clk[1] ? (double)clk[0] / ((double)clk[1] / 100.0) : 0.0;
    // cycles one SIMD spends per wave-instruction: SIMD-cycles available / instructions issued on it
    const double simd_instr = instr * wps;
    r.cyc_per_instr_simd = (double)clk[0] / simd_instr;
    return r;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 16384;
    const char* only = argc > 2 ? argv[2] : nullptr;   // substring filter on the class names (comma-free); all classes without it
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    uint32_t* d_out;
    unsigned long long* d_clk;
    CHECK(hipMalloc((void**)&d_out, (size_t)(2 + 2 * n_cus * 1024) * 4));
    CHECK(hipMemset(d_out, 0x5A, (size_t)(2 + 2 * n_cus * 1024) * 4));
    CHECK(hipMalloc((void**)&d_clk, 16));
    const int wps_list[] = {1, 2, 3, 4, 6, 8};
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz_prop\": %d, \"iters\": %d,\n \"note\": \"G wave64-instructions/s for the whole chip; "
           "cyc = shader cycles one SIMD spends per wave-instruction (block 0's s_memtime span / instructions issued per SIMD); dep = one dependent chain per wave, "
           "ind = 8 independent chains per wave\",\n \"classes\": {\n",
           prop.name, prop.gcnArchName, n_cus, prop.clockRate, iters);
#define ROW(OP)                                                                                                          \
    if (!only || strstr(kOpName[OP], only)) {                                                                                                                    \
        printf("  \"%s\": {", kOpName[OP]);                                                                              \
        for (int d = 0; d < 2; ++d) {                                                                                    \
            printf("\"%s\": {", d ? "ind" : "dep");                                                                      \
            for (size_t i = 0; i < sizeof(wps_list) / sizeof(int); ++i) {                                                \
                const int w = wps_list[i];                                                                               \
                Result r = d ? run(k_issue<OP, false>, w, 64 * kOpInstr[OP], iters, n_cus, d_out, d_clk)                \
                             : run(k_issue<OP, true>, w, 64 * kOpInstr[OP], iters, n_cus, d_out, d_clk);               \
                printf("%s\"%d\": {\"G\": %.1f, \"cyc\": %.2f, \"mhz\": %.0f}", i ? ", " : "", w, r.ginstr_s, r.cyc_per_instr_simd, r.mhz); \
            }                                                                                                            \
            printf("}%s", d ? "" : ", ");                                                                                \
        }                                                                                                                \
        printf("},\n");                                                                                                  \
        fflush(stdout);                                                                                                  \
    }
    ROW(OP_BITOP3) ROW(OP_AND) ROW(OP_ADD) ROW(OP_ADDCO_PAIR) ROW(OP_LSHL_OR) ROW(OP_ALIGNBIT) ROW(OP_ADD3) ROW(OP_OR3) ROW(OP_MAD24) ROW(OP_BCNT)
    ROW(OP_LSHL) ROW(OP_LSHL64) ROW(OP_CNDMASK) ROW(OP_BFREV) ROW(OP_MOV) ROW(OP_ADD_F64) ROW(OP_FMA_F64) ROW(OP_DS_READ_B32) ROW(OP_DS_READ_B64) ROW(OP_LSHL_ADD_U64) ROW(OP_XAD) ROW(OP_CMP_CND)
    ROW(OP_BFE) ROW(OP_AND_SDWA) ROW(OP_ADD_SDWA) ROW(OP_ADDCO) ROW(OP_ADDC) ROW(OP_FFBH) ROW(OP_CMP) ROW(OP_AND_OR) ROW(OP_LSHR) ROW(OP_NOT) ROW(OP_SUB) ROW(OP_MIN) ROW(OP_OR) ROW(OP_XNOR) ROW(OP_BFI) ROW(OP_PERM) ROW(OP_LSHLADD) ROW(OP_PK_ADD16) ROW(OP_PK_SUB16) ROW(OP_PK_LSHL16) ROW(OP_PK_LSHR16) ROW(OP_LSHL_SDWA) ROW(OP_ASHR) ROW(OP_AND_LIT) ROW(OP_XOR) ROW(OP_SUBREV)
    {
        // C++ Myers column: instruction count per column taken from the disassembly is not needed — report columns/s
        printf("  \"%s\": {", kOpName[OP_MYERS2]);
        for (int d = 0; d < 4; ++d) {
            printf("\"%s\": {", d == 3 ? "hybrid_bitop3_u64_add_shift" : d == 2 ? "u64_arith_one_chain" : d ? "two_chains" : "one_chain");
            for (size_t i = 0; i < sizeof(wps_list) / sizeof(int); ++i) {
                const int w = wps_list[i];
                const int cols = 16 * (d == 1 ? 2 : 1);
                Result r = d == 3 ? run(k_myers2h, w, cols, iters / 4, n_cus, d_out, d_clk) : d == 2 ? run(k_myers64, w, cols, iters / 4, n_cus, d_out, d_clk) : d ? run(k_myers<false>, w, cols, iters / 4, n_cus, d_out, d_clk) : run(k_myers<true>, w, cols, iters / 4, n_cus, d_out, d_clk);
                printf("%s\"%d\": {\"G_columns\": %.2f, \"cyc_per_column\": %.2f, \"mhz\": %.0f}", i ? ", " : "", w, r.ginstr_s, r.cyc_per_instr_simd, r.mhz);
            }
            printf("}%s", d == 3 ? "" : ", ");
        }
        printf("}\n");
    }
    printf(" }\n}\n");
    return 0;
}
