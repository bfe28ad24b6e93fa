// Dev tool (round 2): does it matter which register file a 16x16x32 MFMA's operands live in?
// One wave per SIMD (256 threads, 1 block per CU), registers only, 32 independent accumulators.
//   variant 0: A, B in VGPRs, C/D in VGPRs      variant 1: A in AGPRs, B, C/D in VGPRs (conv_rs.hip)
//   variant 2: A, B in VGPRs, C/D in AGPRs      variant 3: as 1 with a ds_read_b128 every 3rd MFMA (operand stream)
// prints cycles per MFMA (s_memtime) - 16 = the matrix pipe's issue rate.
// hipcc --offload-arch=gfx950 -O3 mfma_agpr.hip -o mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256, 1) void k(const u32x4* __restrict__ src, float* out, int iters, unsigned long long* cyc) {
  __shared__ u32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = src[i];
  __syncthreads();
  u32x4 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = src[(threadIdx.x * 12 + i) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = src[(threadIdx.x * 12 + 8 + i) & 4095];
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (V == 1 || V == 3) for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(a[i]));
  if (V == 2) for (int i = 0; i < 32; ++i) asm volatile("" : "+a"(acc[i]));
  const u32x4* lp = lds + (threadIdx.x & 63);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (V == 3 && (i % 3) == 0) { b[(i / 3) & 3] = lp[((it + i) & 15) * 64]; }
      if (V == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(b[i & 3]));
      else if (V == 1 || V == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(a[i & 7]), "v"(b[i & 3]));
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 7]), "v"(b[i & 3]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15");
  float s = 0.f;
  for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 5) *cyc = t1 - t0;
}
template <int V> void run(const u32x4* src, float* out, unsigned long long* cyc, const char* what) {
  const int iters = 20000;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, src, out, iters, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, src, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %5.2f cycles per MFMA, %6.0f TFLOP/s\n", what, (double)c / (iters * 32.0), 256.0 * 4 * iters * 32 * 16384.0 / ms / 1e9);
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83FF83FFu) | 0x38003800u; }
  u32x4* src; hipMalloc(&src, h.size() * 4); hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  for (int r = 0; r < 2; ++r) {
    run<0>(src, out, cyc, "A,B VGPR; C/D VGPR");
    run<1>(src, out, cyc, "A AGPR; B, C/D VGPR");
    run<2>(src, out, cyc, "A,B VGPR; C/D AGPR");
    run<3>(src, out, cyc, "A AGPR; B from LDS every 3rd MFMA; C/D VGPR");
  }
  return 0;
}
