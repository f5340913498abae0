// Issue-rate micro-benchmark of the gfx950 vector unit (diagnostics; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip && build/ubench_valu
// For each instruction: W waves per SIMD on every SIMD of the chip run a loop of 64 independent-enough copies of the
// instruction (8 register chains); reported are wave-instructions per second chip-wide, shader cycles per instruction
// per SIMD (s_memtime) and the sustained shader clock (s_memtime ticks per s_memrealtime tick of 100 MHz).
// It answers what DESIGN.md used to infer: how many wave64 instructions per second the chip can issue, whether the
// packed fp32 forms (v_pk_mul_f32 / v_pk_add_f32) issue at the same rate as the scalar ones (= twice the work per
// slot), and what the float<->double conversions of the bilateral filter cost.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

struct Result {
    unsigned long long cycles, ticks;
};

#define KERNEL(NAME, TYPE, INIT, ASM)                                                                           \
    __global__ __launch_bounds__(64) void NAME(Result *out, int iters, float seed) {                           \
        TYPE r0 = INIT(seed, 0), r1 = INIT(seed, 1), r2 = INIT(seed, 2), r3 = INIT(seed, 3), r4 = INIT(seed, 4), \
             r5 = INIT(seed, 5), r6 = INIT(seed, 6), r7 = INIT(seed, 7);                                        \
        TYPE b = INIT(seed, 9);                                                                                 \
        const unsigned long long c0 = __builtin_readcyclecounter();                                             \
        const unsigned long long t0 = wall_clock64();                                                           \
        for (int i = 0; i < iters; i++) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; u++) {                                                     \
                asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                            \
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)   \
                             : "v"(b));                                                                         \
            }                                                                                                   \
        }                                                                                                       \
        const unsigned long long c1 = __builtin_readcyclecounter();                                             \
        const unsigned long long t1 = wall_clock64();                                                           \
        TYPE s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                                         \
        if (threadIdx.x == 0) {                                                                                 \
            out[blockIdx.x].cycles = c1 - c0;                                                                   \
            out[blockIdx.x].ticks = t1 - t0;                                                                    \
        }                                                                                                       \
        if (*(float *)&s == 123.456f) out[blockIdx.x].cycles = 0; /* keep the chains alive */                   \
    }

#define INITF(s, i) ((s) + 0.001f * (i))
#define INITF2(s, i) (f2{(s) + 0.001f * (i), (s) - 0.002f * (i)})
#define INITD(s, i) ((double)(s) + 0.001 * (i))

#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define A_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define A_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %8\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
// float -> double -> float round trip: two instructions per copy
#define A_CVT64(i) "v_cvt_f64_f32 %" #i ", %" #i "\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %8\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"

#define A_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define A_MUL64E(i) "v_mul_f32_e64 %" #i ", %" #i ", %8\n"
#define A_MULABS(i) "v_mul_f32_e64 %" #i ", |%" #i "|, %8\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %8, %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_CVTFU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define A_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_SAD(i) "v_sad_u16 %" #i ", %" #i ", %8, %8\n"
#define A_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define A_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %8\n"
#define A_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define A_CNDV(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define A_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMPS(i) "v_cmp_lt_f32_e64 s[10:11], %" #i ", %8\n"
#define A_CMPX(i) "v_cmp_class_f32 vcc, %" #i ", %8\n"
#define A_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define A_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %8\n"
#define A_MULADD(i) "v_mul_f32 %" #i ", %" #i ", %8\nv_add_f32 %" #i ", %" #i ", %8\n"
#define A_MULFMA(i) "v_mul_f32 %" #i ", %" #i ", %8\nv_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define A_MULSGPR(i) "v_mul_f32 %" #i ", s12, %" #i "\n"
#define A_MULLIT(i) "v_mul_f32 %" #i ", 0x3f8ccccd, %" #i "\n"
#define A_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_READLANE(i) "v_readfirstlane_b32 s12, %" #i "\n"
#define A_CVTF64(i) "v_cvt_f32_f64 %" #i ", %" #i "\n"
// one compare feeding eight selects through vcc (what a compiler emits for several selects on one condition), and the
// same selects with vcc written once before the loop by a scalar move
#define A_CND1(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP1CND8_0(i) "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\n"

// v_cndmask_b32 reading vcc: how far behind the compare may it sit?  Each block: one compare, K unrelated adds, one select.
#define CND_GAP_KERNEL(NAME, BLOCK)                                                                             \
    __global__ __launch_bounds__(64) void NAME(Result *out, int iters, float seed) {                           \
        float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7, b = seed + 9; \
        const unsigned long long c0 = __builtin_readcyclecounter();                                             \
        const unsigned long long t0 = wall_clock64();                                                           \
        for (int i = 0; i < iters; i++) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; u++)                                                       \
                asm volatile(BLOCK : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "vcc", "s10", "s11", "v100", "v101", "v102", "v103", "v104", "v105"); \
        }                                                                                                       \
        const unsigned long long c1 = __builtin_readcyclecounter();                                             \
        const unsigned long long t1 = wall_clock64();                                                           \
        float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                                        \
        if (threadIdx.x == 0) { out[blockIdx.x].cycles = c1 - c0; out[blockIdx.x].ticks = t1 - t0; }           \
        if (s == 123.456f) out[blockIdx.x].cycles = 0;                                                          \
    }
// dependent chains (one register feeding the next instruction): latency, not throughput
CND_GAP_KERNEL(k_dep_bilateral, "v_cvt_f64_f32 v[100:101], %0\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_cvt_f32_f64 %0, v[100:101]\nv_cvt_f64_f32 v[100:101], %0\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_cvt_f32_f64 %0, v[100:101]\n")
CND_GAP_KERNEL(k_dep_add, "v_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\nv_add_f32 %0, %0, %8\n")
CND_GAP_KERNEL(k_dep_fma, "v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\nv_fma_f32 %0, %0, %8, %8\n")
CND_GAP_KERNEL(k_dep_fma64, "v_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\nv_fma_f64 v[100:101], v[102:103], v[104:105], v[100:101]\n")
CND_GAP_KERNEL(k_dep_cvt, "v_cvt_f64_f32 v[100:101], %0\nv_cvt_f32_f64 %0, v[100:101]\nv_cvt_f64_f32 v[100:101], %0\nv_cvt_f32_f64 %0, v[100:101]\nv_cvt_f64_f32 v[100:101], %0\nv_cvt_f32_f64 %0, v[100:101]\n")
CND_GAP_KERNEL(k_dep_rcp, "v_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\n")
CND_GAP_KERNEL(k_cnd_gap0, "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n")
CND_GAP_KERNEL(k_cnd_gap1, "v_cmp_lt_f32 vcc, %0, %8\nv_add_f32 %2, %2, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n")
CND_GAP_KERNEL(k_cnd_gap2, "v_cmp_lt_f32 vcc, %0, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n")
CND_GAP_KERNEL(k_cnd_gap4, "v_cmp_lt_f32 vcc, %0, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_cndmask_b32 %1, %1, %8, vcc\n")
CND_GAP_KERNEL(k_cnd_2nd, "v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n")
CND_GAP_KERNEL(k_cnd_e64_vcc, "v_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\nv_cndmask_b32_e64 %3, %3, %8, vcc\nv_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\n")
CND_GAP_KERNEL(k_cnd_e64_gap4, "v_cmp_lt_f32_e64 s[10:11], %0, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_cndmask_b32_e64 %1, %1, %8, s[10:11]\n")
CND_GAP_KERNEL(k_cnd_smov, "s_mov_b64 vcc, s[10:11]\nv_cndmask_b32 %1, %1, %8, vcc\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\n")
CND_GAP_KERNEL(k_addc_vcc, "v_addc_co_u32 %0, vcc, %0, %8, vcc\nv_addc_co_u32 %1, vcc, %1, %8, vcc\nv_addc_co_u32 %2, vcc, %2, %8, vcc\nv_addc_co_u32 %3, vcc, %3, %8, vcc\nv_addc_co_u32 %4, vcc, %4, %8, vcc\nv_addc_co_u32 %5, vcc, %5, %8, vcc\n")
CND_GAP_KERNEL(k_divfmas_vcc, "v_div_fmas_f32 %0, %0, %8, %8\nv_div_fmas_f32 %1, %1, %8, %8\nv_div_fmas_f32 %2, %2, %8, %8\nv_div_fmas_f32 %3, %3, %8, %8\nv_div_fmas_f32 %4, %4, %8, %8\nv_div_fmas_f32 %5, %5, %8, %8\n")
KERNEL(k_sub, float, INITF, A_SUB)
__global__ __launch_bounds__(64) void k_cmp1_cnd8(Result *out, int iters, float seed) {
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7, b = seed + 9;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n" A_CND1(0) A_CND1(1) A_CND1(2) A_CND1(3) A_CND1(4) A_CND1(5) A_CND1(6) A_CND1(7)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "vcc");
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (threadIdx.x == 0) { out[blockIdx.x].cycles = c1 - c0; out[blockIdx.x].ticks = t1 - t0; }
    if (s == 123.456f) out[blockIdx.x].cycles = 0;
}
__global__ __launch_bounds__(64) void k_cnd_vcc_smov(Result *out, int iters, float seed) {
    float r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7, b = seed + 9;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long t0 = wall_clock64();
    asm volatile("s_mov_b64 vcc, 0x5555\n" ::: "vcc");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile(A_CND1(0) A_CND1(1) A_CND1(2) A_CND1(3) A_CND1(4) A_CND1(5) A_CND1(6) A_CND1(7)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "vcc");
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (threadIdx.x == 0) { out[blockIdx.x].cycles = c1 - c0; out[blockIdx.x].ticks = t1 - t0; }
    if (s == 123.456f) out[blockIdx.x].cycles = 0;
}
KERNEL(k_max, float, INITF, A_MAX)
KERNEL(k_mul_e64, float, INITF, A_MUL64E)
KERNEL(k_mul_abs, float, INITF, A_MULABS)
KERNEL(k_fmac, float, INITF, A_FMAC)
KERNEL(k_and, float, INITF, A_AND)
KERNEL(k_addu, float, INITF, A_ADDU)
KERNEL(k_lshl, float, INITF, A_LSHL)
KERNEL(k_mov, float, INITF, A_MOV)
KERNEL(k_cvtfu, float, INITF, A_CVTFU)
KERNEL(k_rndne, float, INITF, A_RNDNE)
KERNEL(k_fract, float, INITF, A_FRACT)
KERNEL(k_bfe, float, INITF, A_BFE)
KERNEL(k_mul24, float, INITF, A_MUL24)
KERNEL(k_sad, float, INITF, A_SAD)
KERNEL(k_exp, float, INITF, A_EXP)
KERNEL(k_add3, float, INITF, A_ADD3)
KERNEL(k_min3, float, INITF, A_MIN3)
KERNEL(k_cnd_sgpr, float, INITF, A_CNDS)
KERNEL(k_cnd_swapped, float, INITF, A_CNDV)
KERNEL(k_cmp_cnd, float, INITF, A_CMPCND)
KERNEL(k_cmp_sgpr, float, INITF, A_CMPS)
KERNEL(k_cmp_class, float, INITF, A_CMPX)
KERNEL(k_divscale, float, INITF, A_DIVSCALE)
KERNEL(k_divfmas, float, INITF, A_DIVFMAS)
KERNEL(k_muladd, float, INITF, A_MULADD)
KERNEL(k_mulfma, float, INITF, A_MULFMA)
KERNEL(k_mulsgpr, float, INITF, A_MULSGPR)
KERNEL(k_mullit, float, INITF, A_MULLIT)
KERNEL(k_dpp, float, INITF, A_DPP)
KERNEL(k_readlane, float, INITF, A_READLANE)
KERNEL(k_fma, float, INITF, A_FMA)
KERNEL(k_mul, float, INITF, A_MUL)
KERNEL(k_add, float, INITF, A_ADD)
KERNEL(k_pkfma, f2, INITF2, A_PKFMA)
KERNEL(k_pkmul, f2, INITF2, A_PKMUL)
KERNEL(k_pkadd, f2, INITF2, A_PKADD)
KERNEL(k_rcp, float, INITF, A_RCP)
KERNEL(k_sqrt, float, INITF, A_SQRT)
KERNEL(k_floor, float, INITF, A_FLOOR)
KERNEL(k_med3, float, INITF, A_MED3)
KERNEL(k_cndmask, float, INITF, A_CNDMASK)
KERNEL(k_cvti, float, INITF, A_CVTI)
KERNEL(k_add64, double, INITD, A_ADD64)
KERNEL(k_mul64, double, INITD, A_MUL64)
KERNEL(k_fma64, double, INITD, A_FMA64)
KERNEL(k_mad24, float, INITF, A_MAD24)
KERNEL(k_mullo, float, INITF, A_MULLO)
KERNEL(k_lshladd, float, INITF, A_LSHLADD)
KERNEL(k_cmp, float, INITF, A_CMP)
KERNEL(k_divfixup, float, INITF, A_DIVFIX)

// conversions change the register width, so they get kernels of their own: 8 chains of f32 -> f64 -> f32
__global__ __launch_bounds__(64) void k_cvt_roundtrip(Result *out, int iters, float seed) {
    float r[8];
    for (int i = 0; i < 8; i++) r[i] = seed + 0.001f * i;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(r[j]));
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[j]) : "v"(d[j]));
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i];
    if (threadIdx.x == 0) {
        out[blockIdx.x].cycles = c1 - c0;
        out[blockIdx.x].ticks = t1 - t0;
    }
    if (s == 123.456f) out[blockIdx.x].cycles = 0;
}

// LDS 16-bit reads (the depth look-ups of integrate, the taps of the bilateral filter)
__global__ __launch_bounds__(64) void k_ds_read_u16(Result *out, int iters, float seed) {
    __shared__ unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a[8];
    for (int i = 0; i < 8; i++) a[i] = (threadIdx.x * 2 + i * 130 + (unsigned)seed) & 8191u;
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = lds[a[j] & 8191u];
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long t1 = wall_clock64();
    unsigned s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (threadIdx.x == 0) {
        out[blockIdx.x].cycles = c1 - c0;
        out[blockIdx.x].ticks = t1 - t0;
    }
    if (s == 0xdeadbeefu) out[blockIdx.x].cycles = 0;
}

typedef void (*kern_t)(Result *, int, float);
struct Case {
    const char *name;
    kern_t k;
    int per_iter;  // instructions of the kind per loop iteration
};

int main(int argc, char **argv) {
    int cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    cus = prop.multiProcessorCount;
    const int simds = cus * 4;
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    Case cases[] = {
        {"v_fma_f32", k_fma, 64},       {"v_mul_f32", k_mul, 64},         {"v_add_f32", k_add, 64},
        {"v_pk_fma_f32", k_pkfma, 64},  {"v_pk_mul_f32", k_pkmul, 64},    {"v_pk_add_f32", k_pkadd, 64},
        {"v_rcp_f32", k_rcp, 64},       {"v_sqrt_f32", k_sqrt, 64},       {"v_floor_f32", k_floor, 64},
        {"v_med3_f32", k_med3, 64},     {"v_cndmask_b32", k_cndmask, 64}, {"v_cvt_i32_f32", k_cvti, 64},
        {"v_cmp_lt_f32", k_cmp, 64},    {"v_div_fixup_f32", k_divfixup, 64},
        {"v_mad_u32_u24", k_mad24, 64}, {"v_mul_lo_u32", k_mullo, 64},    {"v_lshl_add_u32", k_lshladd, 64},
        {"v_sub_f32", k_sub, 64}, {"v_max_f32", k_max, 64}, {"v_mul_f32_e64 (VOP3)", k_mul_e64, 64}, {"v_mul_f32 |abs| (VOP3)", k_mul_abs, 64},
        {"v_fmac_f32", k_fmac, 64}, {"v_and_b32", k_and, 64}, {"v_add_u32", k_addu, 64}, {"v_lshlrev_b32", k_lshl, 64}, {"v_mov_b32", k_mov, 64},
        {"v_cvt_f32_u32", k_cvtfu, 64}, {"v_rndne_f32", k_rndne, 64}, {"v_fract_f32", k_fract, 64}, {"v_bfe_u32", k_bfe, 64},
        {"v_mul_u32_u24", k_mul24, 64}, {"v_sad_u16", k_sad, 64}, {"v_exp_f32", k_exp, 64}, {"v_add3_u32", k_add3, 64}, {"v_min3_f32", k_min3, 64},
        {"v_cndmask_b32 e64 sgpr mask", k_cnd_sgpr, 64}, {"v_cndmask_b32 vcc (src swapped)", k_cnd_swapped, 64},
        {"v_cmp+v_cndmask (2 instr)", k_cmp_cnd, 128}, {"1 v_cmp + 8 v_cndmask vcc", k_cmp1_cnd8, 72}, {"v_cndmask vcc after s_mov vcc", k_cnd_vcc_smov, 64},
        {"DEP cvt,fma64,cvt chain", k_dep_bilateral, 48}, {"DEP v_add_f32 chain", k_dep_add, 48}, {"DEP v_fma_f32 chain", k_dep_fma, 48},
        {"DEP v_fma_f64 chain", k_dep_fma64, 48}, {"DEP cvt f32<->f64 chain", k_dep_cvt, 48}, {"DEP v_rcp_f32 chain", k_dep_rcp, 48},
        {"cmp,CND,4 add (6 instr)", k_cnd_gap0, 48}, {"cmp,add,CND,3 add", k_cnd_gap1, 48}, {"cmp,2 add,CND,2 add", k_cnd_gap2, 48},
        {"cmp,4 add,CND", k_cnd_gap4, 48}, {"cmp,CND,CND,3 add", k_cnd_2nd, 48}, {"6 v_cndmask_e64 on vcc", k_cnd_e64_vcc, 48},
        {"cmp_e64 s,4 add,CND_e64 s", k_cnd_e64_gap4, 48}, {"s_mov vcc,CND,4 add", k_cnd_smov, 48}, {"6 v_addc_co_u32 (vcc in+out)", k_addc_vcc, 48},
        {"6 v_div_fmas (reads vcc)", k_divfmas_vcc, 48}, {"v_cmp_lt_f32_e64 -> sgpr", k_cmp_sgpr, 64}, {"v_cmp_class_f32", k_cmp_class, 64},
        {"v_div_scale_f32", k_divscale, 64}, {"v_div_fmas_f32", k_divfmas, 64},
        {"v_mul+v_add (2 instr)", k_muladd, 128}, {"v_mul+v_fma (2 instr)", k_mulfma, 128},
        {"v_mul_f32 sgpr operand", k_mulsgpr, 64}, {"v_mul_f32 literal operand", k_mullit, 64}, {"v_add_f32 dpp quad_perm", k_dpp, 64},
        {"v_readfirstlane_b32", k_readlane, 64},
        {"v_add_f64", k_add64, 64},     {"v_mul_f64", k_mul64, 64},       {"v_fma_f64", k_fma64, 64},
        {"cvt f32->f64->f32 (2 instr)", k_cvt_roundtrip, 64},             {"ds_read_u16 (dependent)", k_ds_read_u16, 64},
    };
    Result *d_out;
    const int max_blocks = simds * 8;
    hipMalloc(&d_out, sizeof(Result) * max_blocks);
    std::vector<Result> h(max_blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device: %s, %d CUs, %d SIMDs; %d iterations x 64 instructions per wave\n", prop.gcnArchName, cus, simds, iters);
    printf("%-34s %5s %14s %14s %12s %10s\n", "instruction", "w/SIMD", "Ginstr/s chip", "cyc/instr/SIMD", "clock GHz", "ms");
    for (const Case &c : cases) {
        for (int w : {1, 2, 8}) {
            const int blocks = simds * w;
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, d_out, 16, 1.0f);  // warm-up
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d_out, sizeof(Result) * blocks, hipMemcpyDeviceToHost);
            double cyc = 0, clk = 0;
            for (int i = 0; i < blocks; i++) {
                cyc += (double)h[i].cycles;
                clk += (double)h[i].cycles / (double)h[i].ticks * 0.1;  // GHz: ticks are 100 MHz
            }
            cyc /= blocks;
            clk /= blocks;
            const double n = (double)iters * c.per_iter;
            // cycles per instruction per SIMD: clock * SIMDs / chip-wide rate
            const double rate = n * blocks / (ms * 1e-3) / 1e9;
            printf("%-34s %5d %14.1f %14.2f %12.3f %10.3f\n", c.name, w, rate, clk * simds / rate, clk, ms);
        }
    }
    return 0;
}
