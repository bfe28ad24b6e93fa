// Dev tool: what memory traffic beside a dense MFMA loop costs on this part.  Every workgroup (4 waves,
// one per SIMD, two workgroups per CU) runs the registers-only v_mfma_f32_32x32x16_f16 loop of
// mfma_peak.hip on random data and, every `every` MFMAs, issues one wave-level LDS-DMA
// (global_load_lds_dwordx4, 1 KB) from (a) one hot line, (b) a 2 MB window that stays in L2,
// (c) a 2 GB buffer streamed in order (HBM / Infinity Cache).  Prints TFLOP/s, the in-kernel clock
// and the DMA byte rate, i.e. the clock the chip holds per TB/s of each traffic class.
// hipcc --offload-arch=gfx950 -O3 mfma_traffic.hip -o mfma_traffic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Res { unsigned long long cyc, rt; };
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int EVERY>  // one DMA per EVERY MFMAs (0 = none)
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ seed, const char* __restrict__ buf, size_t span,
                                            float* out, int iters, Res* res) {
  extern __shared__ char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 va = seed[(threadIdx.x * 8 + i) & 4095], vb = seed[(threadIdx.x * 8 + 4 + i) & 4095];
    a[i] = *reinterpret_cast<f16x8*>(&va); b[i] = *reinterpret_cast<f16x8*>(&vb);
  }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // this wave's stream: consecutive kilobytes, all waves of the chip interleaved
  const size_t nwaves = (size_t)gridDim.x * 4, me = (size_t)blockIdx.x * 4 + wave;
  size_t pos = me * 1024;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int m = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
          if (EVERY > 0 && (++m % EVERY) == 0) {
            __builtin_amdgcn_sched_barrier(0);
            dma16(buf + (pos & (span - 1)) + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + wave * 4096 + (m & 3) * 1024));
            pos += nwaves * 1024;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    if ((it & 63) == 63) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + smem[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 17) { res->cyc = t1 - t0; res->rt = r1 - r0; }
}
template <int EVERY> void run(const uint4* seed, const char* buf, size_t span, const char* what) {
  float* out; Res* res; hipMalloc(&out, 1 << 24); hipMalloc(&res, sizeof(Res));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 512;
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<EVERY>, dim3(grid), dim3(256), 16384, 0, seed, buf, span, out, iters, res);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<EVERY>, dim3(grid), dim3(256), 16384, 0, seed, buf, span, out, iters, res);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  Res r; hipMemcpy(&r, res, sizeof(Res), hipMemcpyDeviceToHost);
  const double flop = (double)grid * 4 * iters * 8 * 32768.0;
  const double bytes = EVERY ? (double)grid * 4 * iters * 8 / EVERY * 1024.0 : 0.0;
  printf("%-34s 1 KB per %2d MFMAs: %7.0f TFLOP/s  clock %4.0f MHz  DMA %5.2f TB/s  (%.1f B per MFMA)\n", what, EVERY,
         flop / ms / 1e9, (double)r.cyc / r.rt * 100.0, bytes / ms / 1e9, EVERY ? 1024.0 / EVERY : 0.0);
  hipFree(out); hipFree(res);
}
int main() {
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83FF83FFu) | 0x38003800u; }
  uint4* seed; hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  char* big; const size_t BIG = 2048ull << 20; hipMalloc(&big, BIG); hipMemset(big, 0x3c, BIG);
  run<0>(seed, big, 1024, "no traffic");
  run<8>(seed, big, 1024, "hot line (L1)");
  run<4>(seed, big, 1024, "hot line (L1)");
  run<8>(seed, big, 2u << 20, "2 MB window (L2)");
  run<4>(seed, big, 2u << 20, "2 MB window (L2)");
  run<16>(seed, big, BIG, "2 GB stream (HBM)");
  run<8>(seed, big, BIG, "2 GB stream (HBM)");
  run<4>(seed, big, BIG, "2 GB stream (HBM)");
  return 0;
}
