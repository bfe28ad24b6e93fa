// Dev tool (round 2): which fp16 MFMA shape should the conv kernel's inner loop use?
//
// tools/micro/mfma_peak.hip ranked 16x16x32 BELOW 32x32x16 (1.44 vs 1.9 PF, one
// wave per SIMD, registers only, 0.2 s of load) while MI355X_MICROARCH.md ("DVFS give-back" item 7)
// reports 16x16x32 ~1.12-1.15x FASTER on random data at equal cycles.  This bench re-measures the
// way the guide prescribes: operands re-read from LDS with ds_read_b128 at the conv kernel's own
// reuse ratios, TWO waves per SIMD (two 4-wave workgroups per CU), random data, >= 2 s of load before
// timing, interleaved rounds in one process, reporting wave-cycles AND wall AND the in-kernel clock.
//
//   shape A: v_mfma_f32_32x32x16_f16, the round-1 loop: per tap column 6 weight + 6 pixel fragments
//            (12 ds_read_b128) feed 24 MFMAs into acc[2][4] (128 accumulator registers)
//   shape B: v_mfma_f32_16x16x32_f16, same output tile per wave (4 rows x 32 px x 64 cout) and the
//            same LDS bytes per FLOP: per tap column of a 32-channel chunk 12 pixel fragments
//            (6 rows x 2 pixel blocks) + 3 x 4 weight fragments feed 96 MFMAs into acc[4][2][4]
//   REG variants keep the fragments in registers (no LDS reads) for both shapes.
//
// hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Res { unsigned long long cyc, rt; };

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 2) void k(const uint4* __restrict__ seed, float* out, int iters, Res* res) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 48 KB of random fragments
  for (int i = threadIdx.x; i < 49152 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = seed[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const char* rd = smem + lane * 16 + (threadIdx.x >> 6) * 1024;
  float s = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (SHAPE == 0) {
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint4 wf[6], af[6];
    for (int i = 0; i < 6; ++i) { wf[i] = seed[(threadIdx.x * 12 + i) & 4095]; af[i] = seed[(threadIdx.x * 12 + 6 + i) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        if (LDS) {
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            wf[i] = *reinterpret_cast<const uint4*>(rd + ((g * 12 + i) & 31) * 1024 + (it & 3) * 4096 % 16384);
            af[i] = *reinterpret_cast<const uint4*>(rd + ((g * 12 + 6 + i) & 31) * 1024 + (it & 3) * 4096 % 16384);
          }
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) {
          const int nb = m & 1, mb = (m >> 1) & 3, dy = m >> 3;
          acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&wf[dy * 2 + nb]),
                                                               *reinterpret_cast<const f16x8*>(&af[(mb + dy) % 6]), acc[nb][mb], 0, 0, 0);
        }
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  } else {
    f32x4 acc[4][2][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) acc[i][j][c][e] = 0.f;
    uint4 af[6][2], wf[4];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 2; ++j) af[i][j] = seed[(threadIdx.x * 16 + i * 2 + j) & 4095];
    for (int c = 0; c < 4; ++c) wf[c] = seed[(threadIdx.x * 16 + 12 + c) & 4095];
    for (int it = 0; it < iters; ++it) {
      // one iteration = one tap column of TWO 16-channel chunks = 96 MFMAs of 16 KFLOP = the FLOPs of
      // 48 32x32x16 MFMAs; LDS reads: 12 + 12 = 24 (same bytes per FLOP as shape A's 12 per 24)
      if (LDS) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) af[i][j] = *reinterpret_cast<const uint4*>(rd + ((i * 2 + j) & 31) * 1024 + (it & 3) * 4096 % 16384);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (LDS) {
#pragma unroll
          for (int c = 0; c < 4; ++c) wf[c] = *reinterpret_cast<const uint4*>(rd + ((12 + dy * 4 + c) & 31) * 1024 + (it & 3) * 4096 % 16384);
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int c = 0; c < 4; ++c)
              acc[mb][pb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&wf[c]),
                                                                      *reinterpret_cast<const f16x8*>(&af[mb + dy][pb]), acc[mb][pb][c], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int c = 0; c < 4; ++c) for (int e = 0; e < 4; ++e) s += acc[i][j][c][e];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 17) { res->cyc = t1 - t0; res->rt = r1 - r0; }
}

struct Out { double tf, clk, cyc_per_32k; };
template <int SHAPE, bool LDS> Out run(const uint4* seed, float* out, Res* res, double warm_s) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // SHAPE 0: 72 MFMAs of 32 KFLOP per iteration; SHAPE 1: 96 MFMAs of 16 KFLOP (= 48 x 32 K)
  const int iters = SHAPE == 0 ? 4000 : 6000, grid = 512;
  const double flop_per_launch = (double)grid * 4 * iters * (SHAPE == 0 ? 72 : 48) * 32768.0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipEventRecord(e0);
  float ms = 0;
  do {  // keep the chip under this load for warm_s seconds before measuring
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(grid), dim3(256), 49152, 0, seed, out, iters, res);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  } while (ms < warm_s * 1000.0);
  hipEventRecord(e0);
  const int reps = 8;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(grid), dim3(256), 49152, 0, seed, out, iters, res);
  hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  Res r; hipMemcpy(&r, res, sizeof(Res), hipMemcpyDeviceToHost);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return {flop_per_launch * reps / ms / 1e9, (double)r.cyc / r.rt * 100.0, (double)r.cyc / (iters * (SHAPE == 0 ? 72.0 : 48.0))};
}

int main(int argc, char** argv) {
  const double warm = argc > 1 ? atof(argv[1]) : 2.0;
  std::vector<uint32_t> h(4096 * 4);
  uint4* seed; hipMalloc(&seed, h.size() * 4);
  float* out; Res* res; hipMalloc(&out, 1 << 24); hipMalloc(&res, sizeof(Res));
  uint32_t s = 1;
  for (int mode = 1; mode >= 0; --mode) {
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = mode ? ((s & 0x83FF83FFu) | 0x38003800u) : 0u; }
    hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* what = mode ? "random" : "zeros ";
    for (int round = 0; round < (mode ? 3 : 1); ++round) {
      Out a = run<0, true>(seed, out, res, warm), b = run<1, true>(seed, out, res, warm);
      Out c = run<0, false>(seed, out, res, warm), d = run<1, false>(seed, out, res, warm);
      printf("%s round %d | LDS operands: 32x32x16 %6.0f TF %4.0f MHz %5.1f cyc/32K | 16x16x32 %6.0f TF %4.0f MHz %5.1f cyc/32K | ratio %.3f\n",
             what, round, a.tf, a.clk, a.cyc_per_32k, b.tf, b.clk, b.cyc_per_32k, b.tf / a.tf);
      printf("%s round %d | REG operands: 32x32x16 %6.0f TF %4.0f MHz %5.1f cyc/32K | 16x16x32 %6.0f TF %4.0f MHz %5.1f cyc/32K | ratio %.3f\n",
             what, round, c.tf, c.clk, c.cyc_per_32k, d.tf, d.clk, d.cyc_per_32k, d.tf / c.tf);
      fflush(stdout);
    }
  }
  return 0;
}
