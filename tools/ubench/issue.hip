// VALU ISSUE-RATE microbenchmark for gfx950: how many wave64 VALU instructions per cycle does one SIMD issue when 1, 2, 3, 4
// wavefronts share it, per instruction class -- the ceiling the rollout kernels' `SQ_INSTS_VALU / SIMD-cycles` is priced against
// (is the large-batch Go2 kernel at 53 % or at 100 % of the issue limit?).
//
//   hipcc --offload-arch=gfx950 -O3 -o issue issue.hip && ./issue
//
// One workgroup of 4 W wavefronts per CU (the dispatcher deals a workgroup's wavefronts round-robin over the CU's four SIMDs:
// W per SIMD), every wavefront runs `iters` x 64 instructions of the class -- FOUR independent dependence chains per wavefront,
// so that a lone wavefront is not bound by its own result latency -- between two s_memtime reads; reported: cycles per
// instruction PER SIMD = (slowest wavefront's cycles) / (W x instructions per wavefront).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// one "quad" = four independent instructions of the class (operands: four accumulators a, d, e, f; constants b, c)
#define Q_FMA "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
#define Q_MUL "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"
#define Q_MOV "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n"
#define Q_CND "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n"
#define Q_CMP "v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4\n"
#define Q_INT "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_and_b32 %3, %3, %5\n"
#define Q_DPP "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %0, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
#define Q_RDL "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
#define Q_RCP "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
#define Q_SWAP "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %1, %2\n v_permlane16_swap_b32 %3, %0\n"
// ---- round 6: the classes the rollout kernels actually consist of
#define Q_FMAC "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n"                    // VOP2: dst += a * b (three VGPR reads)
#define Q_FMAC2 "v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %5, %5\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %5, %5\n"                   // ... two DISTINCT VGPR reads
#define Q_FMAS "v_fma_f32 %0, %0, s20, %5\n v_fma_f32 %1, %1, s20, %5\n v_fma_f32 %2, %2, s21, %5\n v_fma_f32 %3, %3, s21, %5\n"     // one SGPR source
#define Q_FMACS "v_fmac_f32 %0, s20, %4\n v_fmac_f32 %1, s20, %4\n v_fmac_f32 %2, s21, %5\n v_fmac_f32 %3, s21, %5\n"              // VOP2 with an SGPR source (the dotR sweep)
#define Q_FMAK "v_fma_f32 %0, %0, 2.0, %5\n v_fma_f32 %1, %1, 2.0, %5\n v_fma_f32 %2, %2, 0.5, %5\n v_fma_f32 %3, %3, 0.5, %5\n"     // one inline constant
#define Q_FMA2 "v_fma_f32 %0, %0, %0, %5\n v_fma_f32 %1, %1, %1, %5\n v_fma_f32 %2, %2, %2, %5\n v_fma_f32 %3, %3, %3, %5\n"         // VOP3, two distinct VGPR reads
#define Q_ADD "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_sub_f32 %2, %2, %5\n v_sub_f32 %3, %3, %5\n"
#define Q_MUL3 "v_mul_f32 %0, %1, %4\n v_mul_f32 %1, %2, %4\n v_mul_f32 %2, %3, %5\n v_mul_f32 %3, %0, %5\n"                         // dst != src
#define Q_FMACD "v_fmac_f32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %2, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define Q_WRL "v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 5\n v_writelane_b32 %2, s20, 7\n v_writelane_b32 %3, s21, 9\n"
#define Q_RFL "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n"
#define Q_MOVS "v_mov_b32 %0, s20\n v_mov_b32 %1, s21\n v_mov_b32 %2, s20\n v_mov_b32 %3, s21\n"
#define Q_MINMAX "v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %5\n v_min_f32 %3, %3, %5\n"
#define Q_CMPS "v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cmp_lt_f32 s[20:21], %2, %4\n v_cmp_lt_f32 s[22:23], %3, %4\n"   // VOP3 compare -> SGPR pair
// v_cndmask: the r05 anomaly (23.5 cycles for a pure vcc stream at every W).  Variants: mask in an SGPR pair (VOP3), destination
// different from the sources, alternating with an independent multiply, inline-constant sources (what `c ? 1.f : 0.f` compiles to)
#define Q_CNDS "v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %5, s[22:23]\n v_cndmask_b32 %3, %3, %5, s[22:23]\n"
#define Q_CNDX "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n"       // dst is not a source
#define Q_CNDM "v_cndmask_b32 %0, %0, %4, vcc\n v_mul_f32 %1, %1, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_mul_f32 %3, %3, %4\n"                  // alternating with v_mul
#define Q_CNDK "v_cndmask_b32 %0, 0, 1.0, vcc\n v_cndmask_b32 %1, 0, 1.0, vcc\n v_cndmask_b32 %2, 0, 1.0, vcc\n v_cndmask_b32 %3, 0, 1.0, vcc\n"    // constants
#define Q_CNDD "v_cndmask_b32 %1, %0, %4, vcc\n v_cndmask_b32 %2, %1, %4, vcc\n v_cndmask_b32 %3, %2, %5, vcc\n v_cndmask_b32 %0, %3, %5, vcc\n"     // one dependent chain
#define Q_CNDN "v_cndmask_b32 %0, %0, %4, vcc\n s_nop 0\n v_cndmask_b32 %2, %2, %5, vcc\n s_nop 0\n"                                      // an s_nop between two
#define Q_CNDP "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n"                   // pairs: cnd cnd mul mul
#define Q_CNDE "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %5, vcc\n v_cndmask_b32 %3, %3, %5, s[20:21]\n"   // VOP2 / VOP3 alternating
// runs of R consecutive VOP2 v_cndmask inside blocks of 16 instructions (the rest v_mul_f32): where does the penalty start?
#define B_CND3 "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n "
#define B_CND4 "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n "
#define B_CND6 "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n "
#define B_CND8 "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n "
#define B_CND12 "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n "
// packed fp32 (two fp32 per lane and instruction; operands are aligned VGPR pairs)
#define Q_PKFMA "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
#define Q_PKMUL "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
#define Q_PKADD "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5\n v_pk_add_f32 %3, %3, %5\n"
// scalar unit and LDS next to the vector unit: does a lone wavefront's SALU / DS instruction cost a VALU issue slot?
#define Q_SALU "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_and_b32 s22, s22, s20\n s_or_b32 s23, s23, s21\n"
#define Q_VS "v_fma_f32 %0, %0, %4, %5\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 s21, s21, 1\n"                 // 2 VALU + 2 SALU, alternating
#define Q_NOP "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define Q_NOP1 "v_fma_f32 %0, %0, %4, %5\n s_nop 1\n v_fma_f32 %1, %1, %4, %5\n s_nop 1\n"
// the rollout kernels' dynamic mix (profiles/r04_pmc_unitree_go2_trot.json): 25 % fma, 12 % add, 11 % mul (full-rate fp32), 15.5 % int32,
// 2 % transcendental, 34 % mov / cndmask / cmp / readlane / DPP -- as 16 instructions: 4 fma, 2 add, 2 mul, 2 int, 2 mov, 2 cndmask, 1 dpp,
// 1 readlane (+ one rcp every fourth block)
#define Q_MIXA "v_fma_f32 %0, %0, %4, %5\n v_add_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
#define Q_MIXB "v_fma_f32 %1, %1, %4, %5\n v_mov_b32 %2, %0\n v_cndmask_b32 %3, %3, %4, vcc\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define Q_MIXC "v_fma_f32 %2, %2, %4, %5\n v_add_f32 %3, %3, %5\n v_mul_f32 %0, %0, %4\n v_lshl_add_u32 %1, %1, 1, %4\n"
#define Q_MIXD "v_fma_f32 %3, %3, %4, %5\n v_mov_b32 %0, %2\n v_cndmask_b32 %1, %1, %5, vcc\n v_readlane_b32 s20, %2, 3\n"

#define KERNEL(NAME, BODY16, ...)                                                                                         \
  __global__ void NAME(unsigned long long* out, const float* src, int iters) {                                           \
    float a = src[threadIdx.x & 63], d = a + 1.f, e = a + 2.f, f = a + 3.f, b = 1.0001f, c = 0.5f;                       \
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a), "v"(c) : "vcc");                                                   \
    __syncthreads();                                                                                                     \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                          \
    for (int i = 0; i < iters; i++) {                                                                                    \
      REP4(asm volatile(BODY16 : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c) : __VA_ARGS__);)                      \
    }                                                                                                                    \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                          \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                    \
    if (a + d + e + f == 12345.678f) out[0] = 0;                                                                         \
  }

typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, BODY16, ...)                                                                                        \
  __global__ void NAME(unsigned long long* out, const float* src, int iters) {                                           \
    const float x = src[threadIdx.x & 63];                                                                               \
    f2 a = {x, x + 0.5f}, d = a + 1.f, e = a + 2.f, f = a + 3.f, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};              \
    __syncthreads();                                                                                                     \
    const unsigned long long t0 = __builtin_readcyclecounter();                                                          \
    for (int i = 0; i < iters; i++) {                                                                                    \
      REP4(asm volatile(BODY16 : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c) : __VA_ARGS__);)                      \
    }                                                                                                                    \
    const unsigned long long t1 = __builtin_readcyclecounter();                                                          \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                    \
    if (a.x + d.y + e.x + f.y == 12345.678f) out[0] = 0;                                                                 \
  }

KERNEL(k_fma, Q_FMA Q_FMA Q_FMA Q_FMA, "memory")
KERNEL(k_fmac, Q_FMAC Q_FMAC Q_FMAC Q_FMAC, "memory")
KERNEL(k_fmac2, Q_FMAC2 Q_FMAC2 Q_FMAC2 Q_FMAC2, "memory")
KERNEL(k_fmas, Q_FMAS Q_FMAS Q_FMAS Q_FMAS, "s20", "s21")
KERNEL(k_fmacs, Q_FMACS Q_FMACS Q_FMACS Q_FMACS, "s20", "s21")
KERNEL(k_fmak, Q_FMAK Q_FMAK Q_FMAK Q_FMAK, "memory")
KERNEL(k_fma2, Q_FMA2 Q_FMA2 Q_FMA2 Q_FMA2, "memory")
KERNEL(k_add, Q_ADD Q_ADD Q_ADD Q_ADD, "memory")
KERNEL(k_mul3, Q_MUL3 Q_MUL3 Q_MUL3 Q_MUL3, "memory")
KERNEL(k_fmacd, Q_FMACD Q_FMACD Q_FMACD Q_FMACD, "memory")
KERNEL(k_wrl, Q_WRL Q_WRL Q_WRL Q_WRL, "s20", "s21")
KERNEL(k_rfl, Q_RFL Q_RFL Q_RFL Q_RFL, "s20", "s21", "s22", "s23")
KERNEL(k_movs, Q_MOVS Q_MOVS Q_MOVS Q_MOVS, "s20", "s21")
KERNEL(k_minmax, Q_MINMAX Q_MINMAX Q_MINMAX Q_MINMAX, "memory")
KERNEL(k_cmps, Q_CMPS Q_CMPS Q_CMPS Q_CMPS, "s20", "s21", "s22", "s23")
KERNEL(k_cnds, Q_CNDS Q_CNDS Q_CNDS Q_CNDS, "s20", "s21", "s22", "s23")
KERNEL(k_cndx, Q_CNDX Q_CNDX Q_CNDX Q_CNDX, "memory")
KERNEL(k_cndm, Q_CNDM Q_CNDM Q_CNDM Q_CNDM, "memory")
KERNEL(k_cndk, Q_CNDK Q_CNDK Q_CNDK Q_CNDK, "memory")
KERNEL(k_cndd, Q_CNDD Q_CNDD Q_CNDD Q_CNDD, "memory")
KERNEL(k_cndr3, B_CND3, "memory")
KERNEL(k_cndr4, B_CND4, "memory")
KERNEL(k_cndr6, B_CND6, "memory")
KERNEL(k_cndr8, B_CND8, "memory")
KERNEL(k_cndr12, B_CND12, "memory")
KERNEL(k_cndn, Q_CNDN Q_CNDN Q_CNDN Q_CNDN, "memory")
KERNEL(k_cndp, Q_CNDP Q_CNDP Q_CNDP Q_CNDP, "memory")
KERNEL(k_cnde, Q_CNDE Q_CNDE Q_CNDE Q_CNDE, "s20", "s21")
KERNEL2(k_pkfma, Q_PKFMA Q_PKFMA Q_PKFMA Q_PKFMA, "memory")
KERNEL2(k_pkmul, Q_PKMUL Q_PKMUL Q_PKMUL Q_PKMUL, "memory")
KERNEL2(k_pkadd, Q_PKADD Q_PKADD Q_PKADD Q_PKADD, "memory")
KERNEL(k_salu, Q_SALU Q_SALU Q_SALU Q_SALU, "s20", "s21", "s22", "s23", "scc")
KERNEL(k_vs, Q_VS Q_VS Q_VS Q_VS, "s20", "s21", "scc")
KERNEL(k_nop, Q_NOP Q_NOP Q_NOP Q_NOP, "memory")
KERNEL(k_nop1, Q_NOP1 Q_NOP1 Q_NOP1 Q_NOP1, "memory")
KERNEL(k_mul, Q_MUL Q_MUL Q_MUL Q_MUL, "memory")
KERNEL(k_mov, Q_MOV Q_MOV Q_MOV Q_MOV, "memory")
KERNEL(k_cnd, Q_CND Q_CND Q_CND Q_CND, "memory")
KERNEL(k_cmp, Q_CMP Q_CMP Q_CMP Q_CMP, "vcc")
KERNEL(k_int, Q_INT Q_INT Q_INT Q_INT, "memory")
KERNEL(k_dpp, Q_DPP Q_DPP Q_DPP Q_DPP, "memory")
KERNEL(k_rdl, Q_RDL Q_RDL Q_RDL Q_RDL, "s20", "s21", "s22", "s23")
KERNEL(k_rcp, Q_RCP Q_RCP Q_RCP Q_RCP, "memory")
KERNEL(k_swap, Q_SWAP Q_SWAP Q_SWAP Q_SWAP, "memory")
KERNEL(k_mix, Q_MIXA Q_MIXB Q_MIXC Q_MIXD, "s20")

struct Case { const char* name; void (*fn)(unsigned long long*, const float*, int); };

int main() {
  const Case cases[] = {{"v_fma_f32 d, d, v, v (VOP3, three VGPR sources; 4 independent chains per wavefront)", k_fma},
                        {"v_fma_f32 d, d, d, v (VOP3, two distinct VGPRs)", k_fma2}, {"v_fma_f32 d, d, s, v (one SGPR source)", k_fmas},
                        {"v_fma_f32 d, d, 2.0, v (one inline constant)", k_fmak}, {"v_fmac_f32 d, v, v (VOP2, three VGPR reads)", k_fmac},
                        {"v_fmac_f32 d, v, v (same VGPR twice)", k_fmac2}, {"v_fmac_f32 d, s, v (VOP2, SGPR source: the M v / J v sweep)", k_fmacs},
                        {"v_fmac_f32_dpp row_newbcast", k_fmacd}, {"v_mul_f32 d, d, v", k_mul}, {"v_mul_f32 d, v', v (dst not a source)", k_mul3},
                        {"v_add_f32 / v_sub_f32", k_add}, {"v_max_f32 / v_min_f32", k_minmax}, {"v_mov_b32 v, v", k_mov}, {"v_mov_b32 v, s", k_movs},
                        {"v_cndmask_b32 d, d, v, vcc (r05: the 23.5-cycle anomaly)", k_cnd}, {"v_cndmask_b32 d, d, v, s[..] (VOP3 mask in SGPRs)", k_cnds},
                        {"v_cndmask_b32 d, v, v, vcc (dst not a source)", k_cndx}, {"v_cndmask_b32 d, 0, 1.0, vcc (constants)", k_cndk},
                        {"v_cndmask_b32 vcc, ONE dependent chain", k_cndd}, {"v_cndmask_b32 vcc alternating with v_mul_f32", k_cndm},
                        {"v_cndmask_b32 vcc alternating with s_nop 0, per instruction", k_cndn}, {"v_cndmask_b32 vcc in PAIRS (cnd cnd mul mul)", k_cndp},
                        {"v_cndmask_b32 VOP2 (vcc) alternating with VOP3 (s[..])", k_cnde},
                        {"runs of 3 VOP2 v_cndmask in blocks of 16 (rest v_mul_f32)", k_cndr3}, {"runs of 4 VOP2 v_cndmask in blocks of 16 (rest v_mul_f32)", k_cndr4}, {"runs of 6 VOP2 v_cndmask in blocks of 16 (rest v_mul_f32)", k_cndr6}, {"runs of 8 VOP2 v_cndmask in blocks of 16 (rest v_mul_f32)", k_cndr8}, {"runs of 12 VOP2 v_cndmask in blocks of 16 (rest v_mul_f32)", k_cndr12},
                        {"v_cmp_lt_f32 -> vcc", k_cmp}, {"v_cmp_lt_f32 -> s[..] (VOP3)", k_cmps}, {"int32 (add / lshl_add / and)", k_int},
                        {"DPP (v_mov_dpp row_shr, v_add_dpp quad_perm)", k_dpp}, {"v_readlane_b32", k_rdl}, {"v_readfirstlane_b32", k_rfl},
                        {"v_writelane_b32 (SGPR spill store)", k_wrl}, {"v_rcp_f32 (transcendental)", k_rcp}, {"v_permlane16_swap_b32", k_swap},
                        {"v_pk_fma_f32 (two fp32 per lane)", k_pkfma}, {"v_pk_mul_f32", k_pkmul}, {"v_pk_add_f32", k_pkadd},
                        {"SALU (s_add / s_and / s_or), per instruction", k_salu}, {"v_fma_f32 alternating with s_add_u32, per instruction", k_vs},
                        {"s_nop 0, per instruction", k_nop}, {"v_fma_f32 alternating with s_nop 1, per instruction", k_nop1},
                        {"rollout-kernel mix (8 fp32 + 2 int + 2 mov + 2 cndmask + dpp + readlane)", k_mix}};
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount, iters = 4000, per_wave = iters * 64;
  unsigned long long* out;
  float* src;
  hipMalloc(&out, sizeof(unsigned long long) * ncu * 16);
  hipMalloc(&src, sizeof(float) * 64);
  std::vector<float> h(64);
  for (int i = 0; i < 64; i++) h[i] = 1.f + 0.001f * i;
  hipMemcpy(src, h.data(), sizeof(float) * 64, hipMemcpyHostToDevice);
  printf("tools/ubench/issue.hip on %s (%d CUs): cycles per wave64 VALU instruction PER SIMD with W wavefronts on the SIMD\n", prop.gcnArchName, ncu);
  printf("(one workgroup of 4 W wavefronts per CU; s_memtime ticks of the slowest wavefront / (W x %d instructions))\n\n", per_wave);
  printf("%-96s %8s %8s %8s %8s\n", "instruction class", "W=1", "W=2", "W=3", "W=4");
  for (const Case& c : cases) {
    printf("%-96s", c.name);
    for (int W = 1; W <= 4; W++) {
      std::vector<unsigned long long> r(ncu * 4 * W);
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(c.fn, dim3(ncu), dim3(64 * 4 * W), 0, 0, out, src, iters);
        hipDeviceSynchronize();
      }
      hipMemcpy(r.data(), out, sizeof(unsigned long long) * r.size(), hipMemcpyDeviceToHost);
      unsigned long long worst = 0;
      for (auto v : r) worst = v > worst ? v : worst;
      printf(" %8.2f", (double)worst / ((double)W * per_wave));
    }
    printf("\n");
  }
  printf("\nReading: a class whose W = 4 figure is 2.0 issues at the full fp32 rate (one wave64 instruction per 2 cycles per SIMD, the\n"
         "157 TFLOP/s vector peak for FMA); 4.0 = half rate (16 lanes per cycle); the W = 1 column is what ONE wavefront can issue on its\n"
         "own with four independent chains.  The mix row is the ceiling `SQ_INSTS_VALU x cycles-per-instruction / SIMD-cycles` is priced at.\n");
  return 0;
}
