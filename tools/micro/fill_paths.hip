// Dev tool: global -> LDS fill throughput of the two paths, 8 waves per CU, no other work:
//   (a) LDS-DMA  global_load_lds_dwordx4 (1 KB per wave-instruction, no VGPRs)
//   (b) global_load_dwordx4 -> VGPR -> ds_write_b128
// from a 2 MB L2-resident window and from a 2 GB stream.
// hipcc --offload-arch=gfx950 -O3 fill_paths.hip -o fill_paths
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int PATH>
__global__ __launch_bounds__(256, 2) void k(const char* __restrict__ buf, size_t span, float* out, int iters) {
  extern __shared__ char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t nwaves = (size_t)gridDim.x * 4, me = (size_t)blockIdx.x * 4 + wave;
  size_t pos = me * 1024;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (PATH == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dma16(buf + (pos & (span - 1)) + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + wave * 8192 + u * 1024));
        pos += nwaves * 1024;
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        v[u] = *reinterpret_cast<const uint4*>(buf + (pos & (span - 1)) + lane * 16);
        pos += nwaves * 1024;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) *reinterpret_cast<uint4*>(smem + wave * 8192 + u * 1024 + lane * 16) = v[u];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc.x = smem[threadIdx.x * 16];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc.x;
}
template <int PATH> void run(const char* buf, size_t span, const char* what) {
  float* out; hipMalloc(&out, 1 << 22);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, grid = 512;
  hipLaunchKernelGGL(k<PATH>, dim3(grid), dim3(256), 32768, 0, buf, span, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<PATH>, dim3(grid), dim3(256), 32768, 0, buf, span, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * 4 * iters * 8 * 1024.0;
  printf("%-44s %6.2f TB/s  (%5.1f GB/s per CU)\n", what, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
  hipFree(out);
}
int main() {
  char* big; const size_t BIG = 2048ull << 20; hipMalloc(&big, BIG); hipMemset(big, 0x3c, BIG);
  run<0>(big, 2u << 20, "LDS-DMA, 2 MB window (L2)");
  run<1>(big, 2u << 20, "load -> VGPR -> ds_write, 2 MB window (L2)");
  run<0>(big, BIG, "LDS-DMA, 2 GB stream");
  run<1>(big, BIG, "load -> VGPR -> ds_write, 2 GB stream");
  run<0>(big, 64u << 10, "LDS-DMA, 64 KB window (L1/L2)");
  run<1>(big, 64u << 10, "load -> VGPR -> ds_write, 64 KB window");
  return 0;
}
