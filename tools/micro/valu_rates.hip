// Dev tool: issue cost of the vector instructions FSRCNN's matrix-core kernels are made of, in cycles per wave64 instruction, measured as
// 4 waves per SIMD x long dependent-free streams (four independent chains per wave).  Since the matrix pipe and the vector ALU of a SIMD
// do not overlap on this part (mfma_valu_phases.hip), these are the prices a kernel pays per instruction next to its MFMAs.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize valu_rates.hip -o valu_rates ; ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ float row_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true)); }

// KIND: 0 v_add_f32, 1 v_add_f32_dpp wave_shl:1, 2 v_add_f32_dpp row_shr:1, 3 v_pk_mul_f16, 4 v_pk_max_f16, 5 v_cvt_pk_f16_f32 (RNE),
//       6 v_cvt_pkrtz_f16_f32, 7 v_fma_f32, 8 v_max_f32, 9 v_mov_b64 (pair copy), 10 v_pk_add_f32, 11 v_fma_mix_f32 (f16 operand)
template <int KIND>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int iters) {
  float c[4];
  for (int i = 0; i < 4; ++i) c[i] = src[(threadIdx.x + 64 * i) & 1023];
  h2 hv[4];
  for (int i = 0; i < 4; ++i) hv[i] = h2{(_Float16)c[i], (_Float16)(c[i] * 0.5f)};
  const h2 hs = {(_Float16)0.99f, (_Float16)1.01f};
  f2 pv[4];
  for (int i = 0; i < 4; ++i) pv[i] = f2{c[i], c[i] + 1.f};
  unsigned u[4] = {1u, 2u, 3u, 4u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (KIND == 0) c[i] = c[i] + 1.0009765625f;
        if constexpr (KIND == 1) c[i] = wave_shl1(c[i]) + c[(i + 1) & 3];
        if constexpr (KIND == 2) c[i] = row_shr1(c[i]) + c[(i + 1) & 3];
        if constexpr (KIND == 3) hv[i] = hv[i] * hs;
        if constexpr (KIND == 4) hv[i] = __builtin_elementwise_max(hv[i], hv[(i + 1) & 3]);
        // (a conversion needs a consumer and a changing input: + v_xor_b32 + v_add_f32 per conversion, three instructions counted)
        if constexpr (KIND == 5) { const f2 t = {c[i], c[(i + 1) & 3]}; const h2 q = __builtin_convertvector(t, h2); u[i] ^= __builtin_bit_cast(unsigned, q); c[i] += 1.0009765625f; }
        if constexpr (KIND == 6) { u[i] ^= __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(c[i], c[(i + 1) & 3])); c[i] += 1.0009765625f; }
        if constexpr (KIND == 7) c[i] = __builtin_fmaf(c[i], 0.999f, 0.001f);
        if constexpr (KIND == 8) c[i] = __builtin_fmaxf(c[i], c[(i + 1) & 3]);
        if constexpr (KIND == 9) { asm volatile("v_mov_b64 %0, %1" : "=v"(pv[i]) : "v"(pv[(i + 1) & 3])); }
        if constexpr (KIND == 10) pv[i] = pv[i] + pv[(i + 1) & 3];
        if constexpr (KIND == 11) c[i] = __builtin_fmaf((float)hv[i][0], -2048.f, c[i]);
        if constexpr (KIND == 12) c[i] = c[i] + c[(i + 1) & 3];                       // v_add_f32 with the operand pattern of kinds 1, 2, 4, 8
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i] + (float)hv[i][0] + (float)hv[i][1] + pv[i][0] + pv[i][1] + (float)u[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* what, const float* src, float* out, int cus, int per_iter) {
  const int iters = 20000, W = 4;
  hipLaunchKernelGGL(k<KIND>, dim3(cus * W), dim3(256), 0, 0, src, out, 100);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(cus * W), dim3(256), 0, 0, src, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: W waves x iters x per_iter instructions in ms
  const double ns_per_instr = ms * 1e6 / ((double)W * iters * per_iter);
  std::printf("%-44s %6.3f ns per wave-instruction = %5.2f cycles at 2.1 GHz (%d counted per iteration; see the ISA for the exact count)\n", what, ns_per_instr,
              ns_per_instr * 2.1, per_iter);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* src; float* out;
  (void)hipMalloc(&src, 1024 * 4); (void)hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1.f + (float)i / 1024.f;
  (void)hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  run<0>("v_add_f32", src, out, cus, 64);
  run<1>("v_add_f32_dpp wave_shl:1", src, out, cus, 64);
  run<2>("v_add_f32_dpp row_shr:1", src, out, cus, 64);
  run<3>("v_pk_mul_f16", src, out, cus, 64);
  run<4>("v_pk_max_f16", src, out, cus, 64);
  run<5>("v_cvt_pk_f16_f32 + v_xor_b32 + v_add_f32", src, out, cus, 192);
  run<6>("v_cvt_pkrtz_f16_f32 + v_xor_b32 + v_add_f32", src, out, cus, 192);
  run<7>("v_fma_f32", src, out, cus, 64);
  run<8>("v_max_f32", src, out, cus, 64);
  run<9>("v_mov_b64", src, out, cus, 64);
  run<10>("v_pk_add_f32", src, out, cus, 64);
  run<11>("v_fma_mix_f32", src, out, cus, 64);
  run<12>("v_add_f32, operand from the next chain", src, out, cus, 64);
  return 0;
}
