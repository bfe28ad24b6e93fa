// Dev tool (round 5): instruction-mix skeletons of a Winograd-domain 3x3 conv tile for a 64-cout fp16 layer, to be timed at
// the 1400 W cap beside the real direct kernel (conv_w16.hip via ss4k_bench_conv) BEFORE any real kernel is written.  No correct addressing:
// every LDS read / LDS-DMA / store / VALU op / MFMA of the inner loop and of the epilogue is there in the right count and with real data
// dependencies (MFMA operands are computed from what the LDS reads return), on random fp16 data.
//
//   W2D: F(2x2,3x3).  One wave per SIMD, a wave owns 32 couts x 32 tiles (2x2 outputs each), all 16 Winograd positions in 256 accumulator
//        registers.  Per K-chunk (16 cin) and wave: 16 ds_read_b128 of U (transformed weights, LDS-DMA'd per chunk), the wave's own input
//        transform IN REGISTERS in MFMA B-operand layout (16 ds_read_b128 of its tiles' 4x4 patches, 128 v_pk_add_f16), 16 MFMA 32x32x16.
//        2.25 x fewer matrix ops than direct convolution; 2 LDS reads + 8 packed VALU ops per MFMA.
//   W1D: F(2,3) along x, direct along y.  One wave per SIMD, a wave owns 32 couts x 4 rows x 32 column pairs, 4 positions x 4 rows = 256
//        accumulator registers.  Per K-chunk and wave: 12 ds_read_b128 of U, per input row (6) 4 ds_read_b128 + 16 v_pk ops of transform,
//        48 MFMA 32x32x16.  1.5 x fewer matrix ops; 0.75 LDS reads + 2 packed VALU ops per MFMA.
//   W1D2: the same with TWO waves per SIMD: a wave owns 32 couts x 2 rows x 32 column pairs (4 positions x 2 rows = 128 accumulator registers),
//        two workgroups per CU; per K-chunk and wave 12 reads of U + 4 input rows x (4 reads + 16 v_pk ops), 24 MFMAs (1.17 reads per MFMA).
// Both: LDS-DMA of the next chunk's halo tile + U through the MFMA stream (L2-resident window), one barrier per chunk, and every NCH chunks
// the inverse transform + LeakyReLU + fp16 pack + non-temporal stores of the tile.
// Reported: direct-equivalent TFLOP/s = output pixels x 64 couts x 2 x 9 x Cin / time (what conv_roofline / ss4k_bench_conv report for the
// direct kernel), and the MFMA-pipe TFLOP/s actually issued.
// hipcc --offload-arch=gfx950 -O3 wino_skeleton.hip -o wino_skeleton ; ./wino_skeleton [seconds per variant]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ f16x8 ldsv(const char* p) { return *reinterpret_cast<const f16x8*>(p); }

constexpr int LDS_STAGE = 64 * 1024;   // one stage = a chunk's halo tile + its U (W2D: 10.9 + 32 KB, W1D: 21 + 24.6 KB); two stages
constexpr int LDS_STAGE2 = 38 * 1024;  // W1D2 (two workgroups per CU): 4 + 2 rows x 66 columns = 12.7 KB + U 24.6 KB

// MODE 0 = W2D, 1 = W1D, 2 = W1D2.  NCH = K-chunks per tile (4: a 64 -> 64 layer, 12: conv5 of an RDB)
template <int MODE, int NCH>
__global__ __launch_bounds__(256, MODE == 2 ? 2 : 1) void k(const uint4* __restrict__ seed, const char* __restrict__ buf, char* __restrict__ outp, int tiles) {
  extern __shared__ char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int STAGE = MODE == 2 ? LDS_STAGE2 : LDS_STAGE, ROWS = MODE == 2 ? 2 : 4, NACC = MODE == 2 ? 8 : 16;
  for (int i = threadIdx.x; i < 2 * STAGE / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = seed[i & 4095];
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  size_t pos = ((size_t)blockIdx.x * 4 + wave) * 1024;
  const size_t stride = (size_t)gridDim.x * 4 * 1024, span = (size_t)(2u << 20);
  char* myout = outp + ((size_t)blockIdx.x * 256 + threadIdx.x) * 512;
  constexpr int NDMA = MODE == 0 ? 11 : MODE == 1 ? 12 : 10;   // LDS-DMA wave-instructions (1 KB each) per wave and chunk: (halo tile + U) / 4 waves
  int c = 0;
  for (int t = 0; t < tiles; ++t) {
    for (int kc = 0; kc < NCH; ++kc, ++c) {
      const unsigned stage = lds0 + (c & 1) * STAGE;
      const char* rd = smem + ((c & 1) ^ 1) * STAGE;
      const char* rdU = rd + (MODE == 2 ? 13 : 24) * 1024 + (wave & 1) * (MODE == 2 ? 12288 : 16384) + lane * 16;   // this wave's cout half of U
      const char* rdD = rd + (wave >> 1) * 4096 + lane * 16;                 // this wave's tiles / rows
      int nd = 0;
      auto dma = [&]() {
        if (nd < NDMA) {
          __builtin_amdgcn_sched_barrier(0);
          dma16(buf + (pos & (span - 1)) + lane * 16, stage + ((wave * NDMA + nd) * 1024) % STAGE);
          pos += stride; ++nd;
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if constexpr (MODE == 0) {
        // input transform in registers: the lane's tile (4x4 pixels) x 8 channels -> 16 positions x 8 channels (B operand of position p)
        f16x8 d[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = ldsv(rdD + i * 1024);
        f16x8 r[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // rows: B^T d (per column j)
          r[0 * 4 + i] = d[0 * 4 + i] - d[2 * 4 + i];
          r[1 * 4 + i] = d[1 * 4 + i] + d[2 * 4 + i];
          r[2 * 4 + i] = d[2 * 4 + i] - d[1 * 4 + i];
          r[3 * 4 + i] = d[1 * 4 + i] - d[3 * 4 + i];
        }
        f16x8 v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // columns: (.) B
          v[i * 4 + 0] = r[i * 4 + 0] - r[i * 4 + 2];
          v[i * 4 + 1] = r[i * 4 + 1] + r[i * 4 + 2];
          v[i * 4 + 2] = r[i * 4 + 2] - r[i * 4 + 1];
          v[i * 4 + 3] = r[i * 4 + 1] - r[i * 4 + 3];
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          const f16x8 u = ldsv(rdU + p * 1024);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u, v[p], acc[p], 0, 0, 0);
          if (p >= 2) dma();
        }
      } else {
        f16x8 u[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) u[i] = ldsv(rdU + i * 1024);
#pragma unroll
        for (int r = 0; r < ROWS + 2; ++r) {   // input rows of the wave's output rows
          const f16x8 d0 = ldsv(rdD + (r * 4 + 0) * 1024), d1 = ldsv(rdD + (r * 4 + 1) * 1024), d2 = ldsv(rdD + (r * 4 + 2) * 1024),
                      d3 = ldsv(rdD + (r * 4 + 3) * 1024);
          f16x8 v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int orow = r - dy;
            if (orow < 0 || orow >= ROWS) continue;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              acc[p * ROWS + orow] = __builtin_amdgcn_mfma_f32_32x32x16_f16(u[p * 3 + dy], v[p], acc[p * ROWS + orow], 0, 0, 0);
              if (p == 1 || p == 3) dma();
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // epilogue of the tile: inverse transform, LeakyReLU, fp16, non-temporal stores
    if constexpr (MODE == 0) {
      // out (2x2) = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]: 16 positions -> 4 outputs per (cout, tile)
      f32x16 o[4];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float tr[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tr[0][j] = acc[0 * 4 + j][e] + acc[1 * 4 + j][e] + acc[2 * 4 + j][e];
          tr[1][j] = acc[1 * 4 + j][e] - acc[2 * 4 + j][e] - acc[3 * 4 + j][e];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) { o[i * 2 + 0][e] = tr[i][0] + tr[i][1] + tr[i][2]; o[i * 2 + 1][e] = tr[i][1] - tr[i][2] - tr[i][3]; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint4 st[2]; __half* hh = reinterpret_cast<__half*>(st);
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float tt = o[q][e]; hh[e] = __float2half(fmaxf(tt, 0.2f * tt)); }
        __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&st[0]), reinterpret_cast<u32x4*>(myout + q * 32));
        __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&st[1]), reinterpret_cast<u32x4*>(myout + q * 32 + 16));
      }
    } else {
#pragma unroll
      for (int row = 0; row < ROWS; ++row) {
        f32x16 o0, o1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          o0[e] = acc[0 * ROWS + row][e] + acc[1 * ROWS + row][e] + acc[2 * ROWS + row][e];
          o1[e] = acc[1 * ROWS + row][e] - acc[2 * ROWS + row][e] - acc[3 * ROWS + row][e];
        }
        uint4 st[4]; __half* hh = reinterpret_cast<__half*>(st);
#pragma unroll
        for (int e = 0; e < 16; ++e) { hh[e] = __float2half(fmaxf(o0[e], 0.2f * o0[e])); hh[16 + e] = __float2half(fmaxf(o1[e], 0.2f * o1[e])); }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&st[q]), reinterpret_cast<u32x4*>(myout + row * 64 + q * 16));
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  }
}

__global__ void k_fill(uint32_t* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t s = (uint32_t)i * 2654435761u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = (s & 0x83FF83FFu) | 0x38003800u;  // two random halves in +-[0.5,1)
  }
}

template <int MODE, int NCH> void run(const uint4* seed, const char* buf, char* out, const char* what, double seconds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = MODE == 2 ? 512 : 256, tiles = MODE == 0 ? 3000 / NCH * 4 : 1000 / NCH * 4;
  const int lds = 2 * (MODE == 2 ? LDS_STAGE2 : LDS_STAGE);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  auto launch = [&]() { hipLaunchKernelGGL((k<MODE, NCH>), dim3(grid), dim3(256), lds, 0, seed, buf, out, tiles); };
  launch(); hipDeviceSynchronize();
  // run for `seconds` so that the chip settles at its power-capped clock, time the last launches
  int n = 0; float ms = 0, total = 0;
  while (total < seconds * 1000.f) {
    hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); total += ms; ++n;
  }
  ms /= 4;
  // per wave and tile: W2D 32 couts x 32 tiles x 4 px; W1D 32 couts x 4 rows x 64 px
  const double px_cout = (MODE == 0 ? 32.0 * 128 : MODE == 1 ? 32.0 * 256 : 32.0 * 128) * 4 /*waves*/ * grid * tiles;
  const double direct_flops = px_cout * 2 * 9 * 16.0 * NCH;
  const double mfma_flops = (MODE == 0 ? 16.0 : MODE == 1 ? 48.0 : 24.0) * NCH * 32768.0 * 4 * grid * tiles;
  printf("%-78s %7.0f direct-equivalent TFLOP/s  (matrix pipe: %5.0f TFLOP/s issued, %.2f ms per launch)\n", what, direct_flops / ms / 1e9, mfma_flops / ms / 1e9, ms);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  std::vector<uint32_t> h(4096 * 4);
  uint32_t s = 1;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s & 0x83FF83FFu) | 0x38003800u; }
  uint4* seed; hipMalloc(&seed, h.size() * 4); hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  char* big; const size_t BIG = 64ull << 20; hipMalloc(&big, BIG); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<uint32_t*>(big), BIG / 4); hipDeviceSynchronize();
  char* out; hipMalloc(&out, 512ull * 256 * 512);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 4>(seed, big, out, "W2D F(2x2,3x3), 64 -> 64 (4 K-chunks per tile), register input transform", seconds);
    run<0, 12>(seed, big, out, "W2D F(2x2,3x3), 192 -> 64 (12 K-chunks per tile: conv5)", seconds);
    run<1, 4>(seed, big, out, "W1D F(2,3) along x, 64 -> 64 (4 K-chunks per tile)", seconds);
    run<1, 12>(seed, big, out, "W1D F(2,3) along x, 192 -> 64 (12 K-chunks per tile: conv5)", seconds);
    run<2, 4>(seed, big, out, "W1D2 F(2,3) along x, two waves per SIMD, 64 -> 64", seconds);
    run<2, 12>(seed, big, out, "W1D2 F(2,3) along x, two waves per SIMD, 192 -> 64 (conv5)", seconds);
  }
  return 0;
}
