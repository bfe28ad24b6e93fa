// Dev tool: do the matrix pipe and the vector ALU of a SIMD overlap when every wave ALTERNATES between a phase of MFMAs and a phase of
// vector instructions that depend on them - the shape of FSRCNN's tail (a row = 14 x v_mfma_f32_32x32x16_f16, then ~ 120 vector
// instructions on their results: profiles/NOTES_r06.md 6)?  Registers only.  One workgroup of 64 W threads per CU-quarter is not
// controllable from HIP, so the grid is one workgroup of 256 threads per CU (one wave per SIMD) times W workgroups per CU.
//   phase A: NM MFMAs as three accumulation chains (as the tail's three tap blocks)
//   phase B: NV vector instructions (v_add_f32 / v_pk_mul_f16 / v_pk_max_f16 mix) that read the accumulators
//   MODE 0: A then B (the tail's order); MODE 1: only A; MODE 2: only B; MODE 3: B of the PREVIOUS iteration's accumulators interleaved
//   by hand between the MFMAs of this one (software pipelining: two accumulator sets)
// hipcc --offload-arch=gfx950 -O3 mfma_valu_phases.hip -o mfma_valu_phases ; ./mfma_valu_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct Res { unsigned long long cyc, rt; };

__device__ __forceinline__ float valu_block(const f32x16& t0, const f32x16& t1, const f32x16& t2, float v) {
  // 48 dependent-on-accumulator vector instructions per call (adds in four independent chains)
  float c0 = v, c1 = v, c2 = v, c3 = v;
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    c0 += t0[i]; c1 += t0[i + 1]; c2 += t0[i + 2]; c3 += t0[i + 3];
    c0 += t1[i]; c1 += t1[i + 1]; c2 += t1[i + 2]; c3 += t1[i + 3];
    c0 += t2[i]; c1 += t2[i + 1]; c2 += t2[i + 2]; c3 += t2[i + 3];
  }
  return (c0 + c1) + (c2 + c3);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, float* out, int iters, Res* res) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 va = src[(threadIdx.x * 8 + i) & 4095], vb = src[(threadIdx.x * 8 + 4 + i) & 4095];
    a[i] = *reinterpret_cast<f16x8*>(&va); b[i] = *reinterpret_cast<f16x8*>(&vb);
  }
  f32x16 z;
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  f32x16 T[3] = {z, z, z}, P[3] = {z, z, z};
  float v = (float)threadIdx.x * 1e-9f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        f32x16 acc = z;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[(s + tb) & 3], acc, 0, 0, 0);
        T[tb] = acc;
      }
      // two more (the expand product) that feed the next iteration's operand
      f32x16 e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], z, 0, 0, 0);
      f32x16 e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[2], z, 0, 0, 0);
      v += e0[0] + e1[0];
    }
    if constexpr (MODE == 4 || MODE == 5) {   // the same matrix work as 28 x v_mfma_f32_16x16x32_f16 (16 cycles each), seven chains of four
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        f32x4 acc = z4;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(s + (c >> 2)) & 3], b[(s + c) & 3], acc, 0, 0, 0);   // (no two chains alike)
#pragma unroll
        for (int e = 0; e < 4; ++e) T[c % 3][(4 * (c / 3) + e) & 15] = acc[e];
      }
    }
    if constexpr (MODE == 0 || MODE == 2 || MODE == 4) {
      if constexpr (MODE == 2) {   // (the accumulators are opaque every iteration: nothing of the block below can be hoisted)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
#pragma unroll
          for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(T[tb][e]));
      }
      // ~ 120 vector instructions on the accumulators: 2.5 blocks of 48
      v = valu_block(T[0], T[1], T[2], v);
      v = valu_block(T[1], T[2], T[0], v);
      float w = 0.f;
#pragma unroll
      for (int i = 0; i < 12; ++i) w += T[2][i] * v;
      v += w;
      if constexpr (MODE == 2) { T[0][0] = v; T[1][1] = v; T[2][2] = v; }
    }
    if constexpr (MODE == 3) {
      // software-pipelined: this iteration's MFMAs into T while the vector block reads P (the previous iteration's), statement by statement
      f32x16 acc0 = z, acc1 = z, acc2 = z;
      float c0 = v, c1 = v, c2 = v, c3 = v;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[s], acc0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { c0 += P[0][4 * s + i]; c1 += P[1][4 * s + i]; }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[(s + 1) & 3], acc1, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { c2 += P[2][4 * s + i]; c3 += P[0][(4 * s + i + 5) & 15]; }
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s], b[(s + 2) & 3], acc2, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { c0 += P[1][(4 * s + i + 3) & 15]; c1 += P[2][(4 * s + i + 7) & 15]; }
      }
      f32x16 e0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], z, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 12; ++i) c2 += P[0][i] * c3;
      f32x16 e1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[2], z, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 12; ++i) c3 += P[1][i] * c2;
      v = (c0 + c1) + (c2 + c3) + e0[0] + e1[0];
      P[0] = acc0; P[1] = acc1; P[2] = acc2;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = v;
  for (int tb = 0; tb < 3; ++tb) for (int e = 0; e < 16; ++e) s += T[tb][e] + P[tb][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { res[blockIdx.x].cyc = t1 - t0; res[blockIdx.x].rt = r1 - r0; }
}

template <int MODE>
static void run(const char* what, const uint4* src, float* out, Res* res, int cus) {
  const int iters = 20000;
  for (int W : {1, 2, 3, 4}) {
    const int blocks = cus * W;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, 200, res);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, iters, res);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<Res> h(blocks);
    (void)hipMemcpy(h.data(), res, sizeof(Res) * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (auto& r : h) { cyc += (double)r.cyc; rt += (double)r.rt; }
    cyc /= blocks; rt /= blocks;
    // s_memtime counts at 100 MHz (constant), so cycles come from the wall time and the shader clock estimate
    const double us_per_iter_per_simd = ms * 1e3 / iters;          // every SIMD runs W waves for `iters` iterations
    std::printf("%-44s W=%d waves/SIMD: %7.3f us per iteration round of W waves = %7.3f us per wave-iteration  (REFCLK ticks/iter %.2f)\n", what, W,
                us_per_iter_per_simd, us_per_iter_per_simd / W, rt / iters);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  uint4* src; float* out; Res* res;
  (void)hipMalloc(&src, 4096 * sizeof(uint4)); (void)hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(float)); (void)hipMalloc(&res, (size_t)cus * 4 * sizeof(Res));
  std::vector<unsigned short> hsrc(4096 * 8);
  for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (unsigned short)(0x2c00 + (i * 2654435761u >> 22 & 0x3ff));   // fp16 in [2^-4, 2^-3)
  (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 2, hipMemcpyHostToDevice);
  std::printf("%d CUs; a wave-iteration = 14 MFMA 32x32x16 (448 matrix cycles) and / or ~ 120 dependent vector instructions (~ 480 issue cycles)\n", cus);
  run<1>("MFMA phase only", src, out, res, cus);
  run<2>("vector phase only", src, out, res, cus);
  run<0>("MFMA phase, then vector phase (the tail)", src, out, res, cus);
  run<3>("software-pipelined by hand", src, out, res, cus);
  run<5>("28 x MFMA 16x16x32 only", src, out, res, cus);
  run<4>("28 x MFMA 16x16x32, then vector phase", src, out, res, cus);
  return 0;
}
