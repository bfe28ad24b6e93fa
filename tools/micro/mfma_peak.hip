// Dev tool: practical MFMA ceiling on this part, to put the conv kernel's TFLOP/s in context.
// Registers-only loops on random data, for both fp16 MFMA shapes, same 64x64 output tile per wave.
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Res { unsigned long long cyc, rt; };
// SHAPE 0: 32x32x16, 2x2 accumulators; SHAPE 1: 16x16x32, 4x4 accumulators (same 64x64 tile, same K=32 per step)
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters, Res* res) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 va = src[(threadIdx.x * 8 + i) & 4095], vb = src[(threadIdx.x * 8 + 4 + i) & 4095];
    a[i] = *reinterpret_cast<f16x8*>(&va); b[i] = *reinterpret_cast<f16x8*>(&vb);
  }
  float s = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 0) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 17) { res->cyc = t1 - t0; res->rt = r1 - r0; }
}
template <int SHAPE> void run(const uint4* src, const char* what) {
  float* out; Res* res; hipMalloc(&out, 1 << 24); hipMalloc(&res, sizeof(Res));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000, grid = 256;
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, src, out, iters, res);  // ~0.2 s of load first
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, src, out, iters, res);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  Res r; hipMemcpy(&r, res, sizeof(Res), hipMemcpyDeviceToHost);
  const double flop = (double)grid * 4 * iters * 8 * 32768.0;
  printf("%s %s: %.2f ms, %.0f TFLOP/s, in-kernel clock %.0f MHz, %.1f cycles per 32x32x16-equivalent\n", what,
         SHAPE ? "16x16x32" : "32x32x16", ms, flop / ms / 1e9, (double)r.cyc / r.rt * 100.0, (double)r.cyc / (iters * 8.0));
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint4* src; hipMalloc(&src, h.size() * 4);
  uint32_t s = 1;
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = mode ? ((s & 0x83FF83FFu) | 0x38003800u) : 0u; }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>(src, mode ? "random" : "zeros "); run<1>(src, mode ? "random" : "zeros ");
    run<0>(src, mode ? "random" : "zeros "); run<1>(src, mode ? "random" : "zeros ");
  }
  return 0;
}
