// Single-wavefront issue / latency microbenchmarks for gfx950 (what does one wave pay per instruction?).
//   hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k(unsigned long long* out, float* sink, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  lds[lane] = (float)lane;
  lds[lane + 64] = 0.f;
  __syncthreads();
  float a = sink[lane], b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f;
  unsigned long long t0, t1;
  int idx = 0;
  // 0: dependent v_fma chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 1: four independent v_fma chains
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                       : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 2: dependent v_readlane -> v_fma (scalar operand) chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("v_readlane_b32 s20, %0, 3\n s_nop 1\n v_fma_f32 %0, s20, %1, %2" : "+v"(a) : "v"(b), "v"(c) : "s20");)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 3: dependent DPP add chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 4: dependent ds_read_b32 chain (address from the loaded value)
  int addr = (lane & 63) * 4;
  float v = 0.f;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("ds_read_b32 %0, %1 offset:256\n s_waitcnt lgkmcnt(0)\n v_cvt_i32_f32 %1, %0\n v_add_u32 %1, %1, %2"
                       : "+v"(v), "+v"(addr) : "v"(lane * 4));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 5: ds_write + fence + ds_read round trip (phase boundary)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(lane * 4));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 6: dependent v_cmp -> v_cndmask chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 7: salu chain
  int s = iters;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) { REP64(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));) }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 8: dependent transcendental chain
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(a));) }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  // 9: ds_read_b128 + wait
  float4 q4;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    REP64(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q4) : "v"((lane & 15) * 16));)
  }
  t1 = __builtin_readcyclecounter();
  if (lane == 0) out[idx] = t1 - t0; idx++;
  sink[lane] = a + d + e + f + v + (float)s + (float)addr + q4.x;
}

int main() {
  const char* names[] = {"dependent v_fma", "4 independent v_fma chains (per instr)", "v_readlane + s_nop 1 + v_fma (per triple)",
                         "s_nop 1 + dependent v_add_dpp (per pair)", "dependent ds_read_b32 + wait + 2 VALU (per round trip)",
                         "ds_write + wait + ds_read + wait", "v_cmp + v_cndmask (per pair)", "dependent s_add", "dependent v_rcp",
                         "ds_read_b128 + wait"};
  const int per[] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
  unsigned long long* out; float* sink;
  hipMalloc(&out, 16 * sizeof(unsigned long long)); hipMalloc(&sink, 64 * sizeof(float));
  hipMemset(sink, 0, 64 * sizeof(float));
  const int iters = 64;
  for (int waves = 1; waves <= 2; waves++) {
    // waves = 2: two workgroups' waves may share a SIMD only by chance; run 1 wave for latency numbers
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, out, sink, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(16);
  hipMemcpy(h.data(), out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  for (int i = 0; i < 10; i++) printf("%-58s %7.2f ticks per unit\n", names[i], (double)h[i] / (iters * per[i] * (i == 1 ? 4 : 1)));
  // wall clock of the counter: time a known-length spin
  return 0;
}
