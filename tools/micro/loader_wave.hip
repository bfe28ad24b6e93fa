// Dev tool: does a dedicated loader wave pay?  Per "chunk" a workgroup runs 72 MFMAs on each of its 4
// compute waves and moves 36 KB global -> LDS (128 B per MFMA, L2-resident source), one barrier per
// chunk, two workgroups per CU:
//   mode 0: the 4 compute waves issue the 36 LDS-DMA instructions themselves (9 each, spread through
//           their MFMA stream) - what conv3x3_kernel does today;
//   mode 1: a fifth wave issues all 36 and waits for them; the compute waves only do MFMAs;
//   mode 2: no traffic (barrier only).
// hipcc --offload-arch=gfx950 -O3 loader_wave.hip -o loader_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 320 : 256, 2) void k(const uint4* __restrict__ seed, const char* __restrict__ buf,
                                                             size_t span, float* out, int chunks) {
  extern __shared__ char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 va = seed[(threadIdx.x * 8 + i) & 4095], vb = seed[(threadIdx.x * 8 + 4 + i) & 4095];
    a[i] = *reinterpret_cast<f16x8*>(&va); b[i] = *reinterpret_cast<f16x8*>(&vb);
  }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  size_t pos = ((size_t)blockIdx.x * 5 + wave) * 1024;
  const size_t stride = (size_t)gridDim.x * 5 * 1024;
  for (int c = 0; c < chunks; ++c) {
    const unsigned stage = lds0 + (c & 1) * 36864;
    if (MODE == 1 && wave == 4) {
#pragma unroll 4
      for (int u = 0; u < 36; ++u) { dma16(buf + (pos & (span - 1)) + lane * 16, stage + u * 1024); pos += stride; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int m = 0; m < 72; ++m) {
        const int ks = (m >> 2) & 1, i = (m >> 1) & 1, j = m & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
        if (MODE == 0 && (m & 7) == 3) {
          __builtin_amdgcn_sched_barrier(0);
          dma16(buf + (pos & (span - 1)) + lane * 16, stage + (wave * 9 + (m >> 3)) * 1024);
          pos += stride;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
}
template <int MODE> void run(const uint4* seed, const char* buf, size_t span, const char* what) {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int chunks = 3000, grid = 512, threads = MODE == 1 ? 320 : 256;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 73728, 0, seed, buf, span, out, chunks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(threads), 73728, 0, seed, buf, span, out, chunks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-52s %7.0f TFLOP/s\n", what, (double)grid * 4 * chunks * 72 * 32768.0 / ms / 1e9);
  hipFree(out);
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83FF83FFu) | 0x38003800u; }
  uint4* seed; hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  char* big; const size_t BIG = 2048ull << 20; hipMalloc(&big, BIG); hipMemset(big, 0x3c, BIG);
  for (int rep = 0; rep < 2; ++rep) {
    run<2>(seed, big, 1024, "no traffic, barrier per chunk");
    run<0>(seed, big, 2u << 20, "compute waves issue the DMAs (L2 window)");
    run<1>(seed, big, 2u << 20, "fifth wave issues the DMAs (L2 window)");
    run<0>(seed, big, BIG, "compute waves issue the DMAs (2 GB stream)");
    run<1>(seed, big, BIG, "fifth wave issues the DMAs (2 GB stream)");
  }
  return 0;
}
