// Would MFMA pay for H = M + J^T D J ?  (north_star: "MFMA is used only if ... recast as a real batched dense GEMM --
// each choice evidenced".)  One wavefront assembles the Go2-sized Newton matrix the way the rollout kernel needs it:
// inputs in LDS (dof-major pyramid rows J^T[i][k], 18 dofs x 16 contact rows; row weights D[k]; M as an 18 x 20
// square), output H in the same square layout, ready for the register Cholesky.
//   variant A: v_mfma_f32_32x32x2_f32, K = 16 -> 8 MFMA instructions on a 32 x 32 tile (18 x 18 used: 32 % of the tile)
//   variant B: the dense VALU formulation (one lane per lower-triangle entry, 16-term dot products from LDS)
// The kernel's own contact-sparse work-list assembly is measured in situ (profiles/r02_sections_*: "H build").
//   hipcc --offload-arch=gfx950 -O3 -o mfma_jtdj mfma_jtdj.hip && ./mfma_jtdj
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NV = 18, K = 16, S = 20;
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(64) bench(unsigned long long* out, float* check, int iters) {
  __shared__ __attribute__((aligned(16))) float Jt[32 * K];   // rows >= NV are zero (tile padding)
  __shared__ __attribute__((aligned(16))) float D[K];
  __shared__ __attribute__((aligned(16))) float M[NV * S];
  __shared__ __attribute__((aligned(16))) float H[NV * S];
  const int lane = threadIdx.x;
  for (int e = lane; e < 32 * K; e += 64) Jt[e] = (e / K) < NV ? 0.01f * (float)((e * 7) % 13) - 0.05f : 0.f;
  for (int e = lane; e < K; e += 64) D[e] = 1.f + 0.1f * e;
  for (int e = lane; e < NV * S; e += 64) M[e] = (e / S == e % S) ? 2.f : 0.01f;
  __syncthreads();
  const int row = lane & 31, half = lane >> 5;
  unsigned long long t0, t1;
  // ---- variant A: MFMA
  t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    float16v acc = {0};
    // one row of J^T per lane (16 floats = 4 x ds_read_b128); lanes >= 32 take the odd k of every pair
    const float4* jr = reinterpret_cast<const float4*>(Jt + row * K);
    const float4* dr = reinterpret_cast<const float4*>(D);
    float jv[K], dv[K];
#pragma unroll
    for (int q = 0; q < K / 4; q++) {
      const float4 a = jr[q], d = dr[q];
      jv[4 * q] = a.x; jv[4 * q + 1] = a.y; jv[4 * q + 2] = a.z; jv[4 * q + 3] = a.w;
      dv[4 * q] = d.x; dv[4 * q + 1] = d.y; dv[4 * q + 2] = d.z; dv[4 * q + 3] = d.w;
    }
#pragma unroll
    for (int k = 0; k < K; k += 2) {
      const float b = half ? jv[k + 1] : jv[k];
      const float a = b * (half ? dv[k + 1] : dv[k]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // acc[i]: row (i / 4) * 8 + half * 4 + i % 4, column `row` (= lane & 31): add M, store the 18 x 18 corner
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = (i / 4) * 8 + half * 4 + (i % 4);
      if (r < NV && row < NV) H[r * S + row] = M[r * S + row] + acc[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D[it & 15] += H[(it % NV) * S + ((it * 5) % NV)] * 1e-9f;   // make the iterations depend on each other through LDS
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[0] = t1 - t0;
  __syncthreads();
  if (lane < NV) check[lane] = H[lane * S + lane];
  // ---- variant B: dense VALU, one lane per lower-triangle entry (171 entries -> 3 passes)
  t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    for (int e = lane; e < NV * (NV + 1) / 2; e += 64) {
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= e) i++;
      const int j = e - i * (i + 1) / 2;
      const float4* ji = reinterpret_cast<const float4*>(Jt + i * K);
      const float4* jj = reinterpret_cast<const float4*>(Jt + j * K);
      const float4* dr = reinterpret_cast<const float4*>(D);
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < K / 4; q++) {
        const float4 a = ji[q], b = jj[q], d = dr[q];
        acc += (a.x * d.x) * b.x + (a.y * d.y) * b.y + (a.z * d.z) * b.z + (a.w * d.w) * b.w;
      }
      const float v = M[i * S + j] + acc;
      H[i * S + j] = v;
      H[j * S + i] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    D[it & 15] += H[(it % NV) * S + ((it * 5) % NV)] * 1e-9f;
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[1] = t1 - t0;
  if (lane < NV) check[NV + lane] = H[lane * S + lane];
}

int main() {
  unsigned long long* out;
  float* check;
  hipMalloc(&out, 2 * sizeof(unsigned long long));
  hipMalloc(&check, 2 * NV * sizeof(float));
  const int iters = 2000;
  hipLaunchKernelGGL(bench, dim3(1), dim3(64), 0, 0, out, check, iters);
  unsigned long long h[2];
  float c[2 * NV];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(c, check, sizeof(c), hipMemcpyDeviceToHost);
  float maxd = 0.f;
  for (int i = 0; i < NV; i++) maxd = fmaxf(maxd, fabsf(c[i] - c[NV + i]) / fabsf(c[i]));
  printf("H = M + J^T D J (Go2: 18 dofs, 16 contact rows), one wavefront, cycles per assembly (incl. LDS in / out):\n");
  printf("  A  v_mfma_f32_32x32x2_f32 x 8                 %8.0f\n", (double)h[0] / iters);
  printf("  B  dense VALU, lane per lower-triangle entry   %8.0f\n", (double)h[1] / iters);
  printf("  diagonal of A vs B: max rel. difference %.2e\n", maxd);
  return 0;
}
