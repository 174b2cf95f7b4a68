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

KERNEL(k_fma, Q_FMA Q_FMA Q_FMA Q_FMA, "memory")
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
  const Case cases[] = {{"v_fma_f32 (4 independent chains per wavefront)", k_fma}, {"v_mul_f32", k_mul}, {"v_mov_b32", k_mov},
                        {"v_cndmask_b32 (vcc)", k_cnd}, {"v_cmp_lt_f32 -> vcc", k_cmp}, {"int32 (add / lshl_add / and)", k_int},
                        {"DPP (v_mov_dpp row_shr, v_add_dpp quad_perm)", k_dpp}, {"v_readlane_b32", k_rdl}, {"v_rcp_f32 (transcendental)", k_rcp},
                        {"v_permlane16_swap_b32", k_swap}, {"rollout-kernel mix (8 fp32 + 2 int + 2 mov + 2 cndmask + dpp + readlane)", k_mix}};
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
  printf("%-78s %8s %8s %8s %8s\n", "instruction class", "W=1", "W=2", "W=3", "W=4");
  for (const Case& c : cases) {
    printf("%-78s", c.name);
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
