// Dev tool: the ingredient ladder of conv3x3_kernel<__half,2,4,4> without its control flow.  Per chunk a
// 4-wave workgroup (two per CU) runs 72 MFMAs per wave; ingredients are switched on one by one:
//   +dma   : the waves issue 38 KB of LDS-DMA per chunk through their MFMA stream (L2 window or 2 GB stream)
//   +lds   : MFMA operands come from LDS (36 ds_read_b128 per wave per chunk, like 6 weight + 6 pixel
//            fragments per tap column) instead of staying in registers
//   +epi   : every 12 chunks (= one conv5 tile) LeakyReLU + fp16 pack + 16 non-temporal 16-byte stores per lane
// hipcc --offload-arch=gfx950 -O3 conv_skeleton.hip -o conv_skeleton
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__global__ void k_fill(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s = (uint32_t)i * 2654435761u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = (s & 0x83FF83FFu) | 0x38003800u;  // two random halves in +-[0.5,1)
  }
}
template <bool DMA, bool LDS, bool EPI>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ seed, const char* __restrict__ buf, size_t span,
                                            char* __restrict__ outp, int chunks, int mix20) {
  extern __shared__ char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 77824 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = seed[i & 4095];
  __syncthreads();
  uint4 wf[6], af[6];
  for (int i = 0; i < 6; ++i) { wf[i] = seed[(threadIdx.x * 12 + i) & 4095]; af[i] = seed[(threadIdx.x * 12 + 6 + i) & 4095]; }
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  size_t pos = ((size_t)blockIdx.x * 4 + wave) * 1024;
  const size_t stride = (size_t)gridDim.x * 4 * 1024;
  char* myout = outp + ((size_t)blockIdx.x * 256 + threadIdx.x) * 256;
  for (int c = 0; c < chunks; ++c) {
    const unsigned stage = lds0 + (c & 1) * 38912;
    const char* rd = smem + ((c & 1) ^ 1) * 38912 + lane * 16;
#pragma unroll
    for (int g = 0; g < 3; ++g) {  // tap column: 6 weight + 6 pixel fragments, 24 MFMAs
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          wf[i] = *reinterpret_cast<const uint4*>(rd + (g * 12 + i) * 1024);
          af[i] = *reinterpret_cast<const uint4*>(rd + (g * 12 + 6 + i) * 1024);
        }
      }
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        const int nb = m & 1, mb = (m >> 1) & 3, dy = m >> 3;
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&wf[dy * 2 + nb]),
                                                             *reinterpret_cast<const f16x8*>(&af[(mb + dy) % 6]), acc[nb][mb], 0, 0, 0);
        if (DMA && (m % 6) == 3 && (g * 4 + m / 6) < 10) {
          __builtin_amdgcn_sched_barrier(0);
          const size_t sp = ((g * 4 + m / 6) * 7 % 20) < mix20 ? span : (size_t)(2u << 20);
          dma16(buf + (pos & (sp - 1)) + lane * 16, stage + (wave * 10 + g * 4 + m / 6) * 1024 % 38912);
          pos += stride;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (EPI && (c % 12) == 11) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          uint4 o[2];
          __half* hh = reinterpret_cast<__half*>(o);
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float t = acc[nb][mb][e]; hh[e] = __float2half(fmaxf(t, 0.2f * t)); acc[nb][mb][e] = 0.f; }
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&o[0]), reinterpret_cast<u32x4*>(myout + (nb * 4 + mb) * 32));
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&o[1]), reinterpret_cast<u32x4*>(myout + (nb * 4 + mb) * 32 + 16));
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 1234.5f) myout[0] = 1;
}
template <bool DMA, bool LDS, bool EPI> void run(const uint4* seed, const char* buf, size_t span, char* out, const char* what, int mix20 = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int chunks = 3000, grid = 512;
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<DMA, LDS, EPI>), dim3(grid), dim3(256), 77824, 0, seed, buf, span, out, chunks, mix20);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<DMA, LDS, EPI>), dim3(grid), dim3(256), 77824, 0, seed, buf, span, out, chunks, mix20);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %7.0f TFLOP/s\n", what, (double)grid * 4 * chunks * 72 * 32768.0 / ms / 1e9);
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83FF83FFu) | 0x38003800u; }
  uint4* seed; hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  char* big; const size_t BIG = 2048ull << 20; hipMalloc(&big, BIG); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(big), BIG / 4); hipDeviceSynchronize();
  char* out; hipMalloc(&out, 512ull * 256 * 256);
  for (int rep = 0; rep < 2; ++rep) {
    run<false, false, false>(seed, big, 1024, out, "MFMA + barrier per chunk");
    run<false, true, false>(seed, big, 1024, out, "+lds operands");
    run<true, false, false>(seed, big, 2u << 20, out, "+dma (L2 window)");
    run<true, true, false>(seed, big, 2u << 20, out, "+dma (L2 window) +lds");
    run<true, true, true>(seed, big, 2u << 20, out, "+dma (L2 window) +lds +epi");
    run<true, true, true>(seed, big, BIG, out, "+dma (45 % of it from the 2 GB stream) +lds +epi", 9);
    run<true, true, true>(seed, big, BIG, out, "+dma (all from the 2 GB stream) +lds +epi", 20);
  }
  return 0;
}
