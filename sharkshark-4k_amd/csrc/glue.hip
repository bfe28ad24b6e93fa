// HBM-bound glue around the networks: every kernel here replaces one torch call of
// FsrcnnUpscalerService.upscale_multi / upscale_single (reference src/upscale/fsrcnn_upscaler.py).
// Image tensors on this side of the networks are fp32 planes (NCHW), exactly the reference's
// layout, so that each op can be parity-tested against its torch counterpart.
#include "common.h"
#include "glue.h"

namespace ss4k {

// every launcher checks its launch: a bad grid surfaces here, not at the next synchronisation
#define SS4K_LAUNCH_OK() SS4K_HIP(hipGetLastError())

static inline dim3 grid1d(size_t n, int block = 256) {
  size_t g = (n + block - 1) / block;
  if (g > 256 * 8 * 4) g = 256 * 8 * 4;  // grid-stride beyond a few waves per CU
  return dim3((unsigned)std::max<size_t>(g, 1));
}

// channel-statistics normalisation, hr = (hr - mean_hr) / (std_hr + 1e-8) * std_lr + mean_lr (fsrcnn_upscaler.py:198-199):
// ONE expression shared by the stand-alone pass and by every kernel that applies it on the fly
struct NormCoef { float mh, sh, ml, sl; };
__device__ __forceinline__ NormCoef norm_coef(const float* __restrict__ st_hr, const float* __restrict__ st_lr, int pl) {
  return NormCoef{st_hr[2 * pl], st_hr[2 * pl + 1] + 1e-8f, st_lr[2 * pl], st_lr[2 * pl + 1]};
}
__device__ __forceinline__ float norm_px(float v, const NormCoef& k) { return (v - k.mh) / k.sh * k.sl + k.ml; }
// The network's HR output tensor is fp32, or fp16 for an fp16 model whose tail can write it (SRVGG: half the bytes of the four
// passes over a 2880 x 5120 tensor): element / 4-element access for both
template <typename HT> __device__ __forceinline__ float hr_ld(const HT* p) { return (float)*p; }
template <typename HT> __device__ __forceinline__ void hr_st(HT* p, float v) { *p = (HT)v; }
template <typename HT> __device__ __forceinline__ float4 hr_ld4(const HT* p);
template <> __device__ __forceinline__ float4 hr_ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 hr_ld4<__half>(const __half* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  const __half2* hh = reinterpret_cast<const __half2*>(&u);
  const float2 a = __half22float2(hh[0]), b = __half22float2(hh[1]);
  return make_float4(a.x, a.y, b.x, b.y);
}
template <typename HT> __device__ __forceinline__ void hr_st4(HT* p, const float4& v);
template <> __device__ __forceinline__ void hr_st4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void hr_st4<__half>(__half* p, const float4& v) {
  uint2 u;
  __half2* hh = reinterpret_cast<__half2*>(&u);
  hh[0] = __floats2half2_rn(v.x, v.y); hh[1] = __floats2half2_rn(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}

// ------------------------------------------------------------------ u8 NHWC -> f32 NCHW (/255)
__global__ void k_u8nhwc_to_f32nchw(const uint8_t* __restrict__ in, float* __restrict__ out, int n, int h,
                                    int w, int c) {
  const size_t hw = (size_t)h * w, total = (size_t)n * hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / hw, p = i - img * hw;
    for (int k = 0; k < c; ++k) out[(img * c + k) * hw + p] = (float)in[i * c + k] / 255.0f;
  }
}
void op_u8nhwc_to_f32nchw(const uint8_t* in, float* out, int n, int h, int w, int c, hipStream_t st) {
  const size_t total = (size_t)n * h * w;
  hipLaunchKernelGGL(k_u8nhwc_to_f32nchw, grid1d(total), dim3(256), 0, st, in, out, n, h, w, c); SS4K_LAUNCH_OK();
}

// ------------------------------------------------------------------ area (adaptive average pool)
__device__ __forceinline__ int a_start(int i, int in, int out) { return (int)floorf((float)(i * in) / out); }
__device__ __forceinline__ int a_end(int i, int in, int out) { return (int)ceilf((float)((i + 1) * in) / out); }

// one block row per output row (blockIdx.y = oy, blockIdx.z = plane): no per-element divisions.
// NORM: the input is read through the channel-statistics normalisation (the normalised tensor is never written)
template <bool NORM, typename HT = float>
__global__ void k_area(const HT* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int oh,
                       int ow, const float* __restrict__ st_hr, const float* __restrict__ st_lr) {
  const int oy = blockIdx.y, pl = blockIdx.z;
  NormCoef nk{};
  if constexpr (NORM) nk = norm_coef(st_hr, st_lr, pl);
  const int y0 = a_start(oy, h, oh), y1 = a_end(oy, h, oh);
  const HT* src = in + (size_t)pl * h * w;
  float* dst = out + ((size_t)pl * oh + oy) * ow;
  for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < ow; ox += gridDim.x * blockDim.x) {
    const int x0 = a_start(ox, w, ow), x1 = a_end(ox, w, ow);
    float sum = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) sum += NORM ? norm_px(hr_ld(src + (size_t)y * w + x), nk) : hr_ld(src + (size_t)y * w + x);
    dst[ox] = sum / (float)(y1 - y0) / (float)(x1 - x0);
  }
}
// Whole-number windows of KX = 4 or 8 columns (the service's HR -> H/8 map, x2 -> lr_shape reductions): the same sums in
// the same order (rows outer, columns inner), the window's columns fetched as 16-byte loads - the scalar form issues KX
// four-byte loads per row whose lanes sit 4*KX bytes apart
template <bool NORM, int KX, typename HT = float>
__global__ void k_area_whole(const HT* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int oh, int ow, int ky,
                             const float* __restrict__ st_hr, const float* __restrict__ st_lr) {
  const int oy = blockIdx.y, pl = blockIdx.z;
  NormCoef nk{};
  if constexpr (NORM) nk = norm_coef(st_hr, st_lr, pl);
  const HT* src = in + ((size_t)pl * h + (size_t)oy * ky) * w;
  float* dst = out + ((size_t)pl * oh + oy) * ow;
  for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < ow; ox += gridDim.x * blockDim.x) {
    float sum = 0.f;
    for (int y = 0; y < ky; ++y) {
      const HT* rp = src + (size_t)y * w + (size_t)ox * KX;
#pragma unroll
      for (int q = 0; q < KX / 4; ++q) {
        const float4 v = hr_ld4<HT>(rp + 4 * q);
        sum += NORM ? norm_px(v.x, nk) : v.x; sum += NORM ? norm_px(v.y, nk) : v.y;
        sum += NORM ? norm_px(v.z, nk) : v.z; sum += NORM ? norm_px(v.w, nk) : v.w;
      }
    }
    dst[ox] = sum / (float)ky / (float)KX;
  }
}
template <bool NORM, typename HT = float>
static bool area_whole(const HT* in, float* out, int planes, int h, int w, int oh, int ow, const float* st_hr, const float* st_lr,
                       hipStream_t st) {
  if (h % oh || w % ow || (reinterpret_cast<uintptr_t>(in) & 15)) return false;
  const int ky = h / oh, kx = w / ow;
  const dim3 g((unsigned)std::min((ow + 255) / 256, 64), (unsigned)oh, (unsigned)planes);
  if (kx == 4) hipLaunchKernelGGL((k_area_whole<NORM, 4, HT>), g, dim3(256), 0, st, in, out, planes, h, w, oh, ow, ky, st_hr, st_lr);
  else if (kx == 8) hipLaunchKernelGGL((k_area_whole<NORM, 8, HT>), g, dim3(256), 0, st, in, out, planes, h, w, oh, ow, ky, st_hr, st_lr);
  else return false;
  return true;
}
static inline dim3 grid_rows(int ow, int oh, int planes) { return dim3((unsigned)std::min((ow + 255) / 256, 64), (unsigned)oh, (unsigned)planes); }
void op_area(const float* in, float* out, int planes, int h, int w, int oh, int ow, hipStream_t st) {
  if (h == oh && w == ow) {
    (void)hipMemcpyAsync(out, in, (size_t)planes * h * w * sizeof(float), hipMemcpyDeviceToDevice, st);
    return;
  }
  SS4K_REQUIRE(oh <= 65535 && planes <= 65535, "area: grid limits");
  if (area_whole<false>(in, out, planes, h, w, oh, ow, nullptr, nullptr, st)) { SS4K_LAUNCH_OK(); return; }
  hipLaunchKernelGGL(k_area<false>, grid_rows(ow, oh, planes), dim3(256), 0, st, in, out, planes, h, w, oh, ow, nullptr, nullptr); SS4K_LAUNCH_OK();
}
template <typename HT>
void op_area_normalized(const HT* in, float* out, int planes, int h, int w, int oh, int ow, const float* st_hr, const float* st_lr,
                        hipStream_t st) {
  SS4K_REQUIRE(oh <= 65535 && planes <= 65535, "area: grid limits");
  if (area_whole<true, HT>(in, out, planes, h, w, oh, ow, st_hr, st_lr, st)) { SS4K_LAUNCH_OK(); return; }
  hipLaunchKernelGGL((k_area<true, HT>), grid_rows(ow, oh, planes), dim3(256), 0, st, in, out, planes, h, w, oh, ow, st_hr, st_lr); SS4K_LAUNCH_OK();
}
template void op_area_normalized<float>(const float*, float*, int, int, int, int, int, const float*, const float*, hipStream_t);
template void op_area_normalized<__half>(const __half*, float*, int, int, int, int, int, const float*, const float*, hipStream_t);

// ------------------------------------------------------------------ per-plane mean / unbiased std
template <typename HT = float>
__global__ void k_stats_partial(const HT* __restrict__ in, double* __restrict__ acc, int hw, int acc_planes = 0, int plane0 = 0) {
  const int pl = blockIdx.y;
  if (acc_planes == 0) acc_planes = gridDim.y;
  const HT* src = in + (size_t)pl * hw;
  double s = 0.0, q = 0.0;
  if ((hw & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    // four values per load, four independent fp64 chains (the sums are order-free: the partials meet in atomics anyway), FOUR loads in
    // flight per thread: with one (rounds 1-5) the pass ran at 4 TB/s - 84 % of its wave cycles waiting (profiles/r06_fsrcnn_f16_sq_counters.json)
    double s4[4] = {0, 0, 0, 0}, q4[4] = {0, 0, 0, 0};
    const size_t n4 = (size_t)hw / 4, stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = hr_ld4<HT>(src + 4 * (i + u * stride));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double d[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { s4[k] += d[k]; q4[k] += d[k] * d[k]; }
      }
    }
    for (; i < n4; i += stride) {
      const float4 v = hr_ld4<HT>(src + 4 * i);
      const double d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { s4[k] += d[k]; q4[k] += d[k] * d[k]; }
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]); q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  } else
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)hw; i += (size_t)gridDim.x * blockDim.x) {
    const double v = hr_ld(src + i);
    s += v; q += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); }
  __shared__ double ss[4], sq[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { ss[wv] = s; sq[wv] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0, Q = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { S += ss[i]; Q += sq[i]; }
    double* a = acc + ((size_t)(blockIdx.x % STATS_SLOTS) * acc_planes + plane0 + pl) * 2;
    atomicAdd(&a[0], S);
    atomicAdd(&a[1], Q);
  }
}
__global__ void k_stats_final(const double* __restrict__ acc, float* __restrict__ stats, int planes, int hw) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= planes) return;
  double S = 0, Q = 0;
  for (int s = 0; s < STATS_SLOTS; ++s) { S += acc[((size_t)s * planes + p) * 2]; Q += acc[((size_t)s * planes + p) * 2 + 1]; }
  const double n = (double)hw;
  const double mean = S / n;
  double var = (Q - S * S / n) / (n - 1.0);  // Bessel-corrected, torch.std default (fsrcnn_upscaler.py:193)
  if (var < 0) var = 0;
  stats[2 * p] = (float)mean;
  stats[2 * p + 1] = (float)sqrt(var);
}
void op_plane_stats_finish(const double* acc, float* stats, int planes, int hw, hipStream_t st) {
  hipLaunchKernelGGL(k_stats_final, dim3((planes + 63) / 64), dim3(64), 0, st, acc, stats, planes, hw); SS4K_LAUNCH_OK();
}
template <typename HT>
void op_plane_stats(double* acc, const HT* in, float* stats, int planes, int hw, hipStream_t st) {
  SS4K_REQUIRE(planes <= STATS_MAX_PLANES, "plane_stats: too many planes");
  SS4K_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 2 * planes * STATS_SLOTS, st));
  int gx = (hw + 256 * 16 - 1) / (256 * 16);
  gx = std::max(1, std::min(gx, 128));
  hipLaunchKernelGGL(k_stats_partial<HT>, dim3(gx, planes), dim3(256), 0, st, in, acc, hw); SS4K_LAUNCH_OK();
  op_plane_stats_finish(acc, stats, planes, hw, st);
}
// the same statistics straight from uint8 NHWC frames (plane 3 f + c = colour c of frame f): every value is (float)byte / 255.0f, the
// conversion kernel's expression, summed in fp64 like k_stats_partial - a job that never materialises its fp32 planes (FSRCNN reading the
// frames itself) still owes the service the low-resolution statistics
__global__ void k_stats_partial_u8(const uint8_t* __restrict__ in, double* __restrict__ acc, int hw, int planes, int plane0 = 0) {
  const int f = blockIdx.y;
  const uint8_t* src = in + (size_t)f * hw * 3;
  __shared__ double lut[256];   // (double)((float)k / 255.0f): the division once per byte value, not once per pixel
  lut[threadIdx.x & 255] = (double)((float)(threadIdx.x & 255) / 255.0f);
  __syncthreads();
  double s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
  // four pixels = twelve bytes = three aligned words per step where the frame allows it
  const bool vec = (hw & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
  if (vec) {
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)hw / 4; i += (size_t)gridDim.x * blockDim.x) {
      const uint32_t w0 = s32[3 * i], w1 = s32[3 * i + 1], w2 = s32[3 * i + 2];
      const uint32_t b[12] = {w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255, w0 >> 24, w1 & 255, (w1 >> 8) & 255, (w1 >> 16) & 255, w1 >> 24,
                              w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24};
#pragma unroll
      for (int k = 0; k < 12; ++k) { const double v = lut[b[k]]; s[k % 3] += v; q[k % 3] += v * v; }
    }
  } else
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)hw; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { const double v = lut[src[3 * i + c]]; s[c] += v; q[c] += v * v; }
  }
  __shared__ double ss[3][4], sq[3][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    for (int off = 32; off > 0; off >>= 1) { s[c] += __shfl_down(s[c], off, 64); q[c] += __shfl_down(q[c], off, 64); }
    if (lane == 0) { ss[c][wv] = s[c]; sq[c][wv] = q[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    double S = 0, Q = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { S += ss[c][i]; Q += sq[c][i]; }
    double* a = acc + ((size_t)(blockIdx.x % STATS_SLOTS) * planes + plane0 + 3 * f + c) * 2;
    atomicAdd(&a[0], S);
    atomicAdd(&a[1], Q);
  }
}
void op_plane_stats_u8nhwc(double* acc, const uint8_t* in, float* stats, int n, int hw, hipStream_t st) {
  const int planes = 3 * n;
  SS4K_REQUIRE(planes <= STATS_MAX_PLANES, "plane_stats: too many planes");
  SS4K_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 2 * planes * STATS_SLOTS, st));
  int gx = (hw + 256 * 16 - 1) / (256 * 16);
  gx = std::max(1, std::min(gx, 128));
  hipLaunchKernelGGL(k_stats_partial_u8, dim3(gx, n), dim3(256), 0, st, in, acc, hw, planes); SS4K_LAUNCH_OK();
  op_plane_stats_finish(acc, stats, planes, hw, st);
}
void op_plane_stats_u8nhwc_partial(double* acc, const uint8_t* in, int n, int hw, int acc_planes, int plane0, hipStream_t st) {
  int gx = (hw + 256 * 16 - 1) / (256 * 16);
  gx = std::max(1, std::min(gx, 128));
  hipLaunchKernelGGL(k_stats_partial_u8, dim3(gx, n), dim3(256), 0, st, in, acc, hw, acc_planes, plane0); SS4K_LAUNCH_OK();
}
template <typename HT>
void op_plane_stats_partial(double* acc, const HT* in, int planes, int hw, int acc_planes, int plane0, hipStream_t st) {
  int gx = (hw + 256 * 16 - 1) / (256 * 16);
  gx = std::max(1, std::min(gx, 128));
  hipLaunchKernelGGL(k_stats_partial<HT>, dim3(gx, planes), dim3(256), 0, st, in, acc, hw, acc_planes, plane0); SS4K_LAUNCH_OK();
}
template void op_plane_stats_partial<float>(double*, const float*, int, int, int, int, hipStream_t);
template void op_plane_stats_partial<__half>(double*, const __half*, int, int, int, int, hipStream_t);
// mean / std of 2 x planes plane records: the first `planes` of hw_a values each -> stats_a, the others of hw_b values -> stats_b
// `rezero`: the partial sums are zeroed as they are read, so that the NEXT job finds clean accumulators without a memset launch
__global__ void k_stats_final2(double* __restrict__ acc, float* __restrict__ sa, float* __restrict__ sb, int planes, int hw_a, int hw_b, int rezero) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= 2 * planes) return;
  double S = 0, Q = 0;
  for (int s = 0; s < STATS_SLOTS; ++s) {
    double2* a = reinterpret_cast<double2*>(acc + ((size_t)s * 2 * planes + p) * 2);
    const double2 v = *a;
    S += v.x; Q += v.y;
    if (rezero) *a = make_double2(0.0, 0.0);
  }
  const double n = (double)(p < planes ? hw_a : hw_b);
  const double mean = S / n;
  double var = (Q - S * S / n) / (n - 1.0);   // (k_stats_final's expressions)
  if (var < 0) var = 0;
  float* o = p < planes ? sa + 2 * p : sb + 2 * (p - planes);
  o[0] = (float)mean;
  o[1] = (float)sqrt(var);
}
void op_plane_stats_finish2(double* acc, float* stats_a, float* stats_b, int planes, int hw_a, int hw_b, bool rezero, hipStream_t st) {
  hipLaunchKernelGGL(k_stats_final2, dim3((2 * planes + 63) / 64), dim3(64), 0, st, acc, stats_a, stats_b, planes, hw_a, hw_b, rezero ? 1 : 0); SS4K_LAUNCH_OK();
}
template void op_plane_stats<float>(double*, const float*, float*, int, int, hipStream_t);
template void op_plane_stats<__half>(double*, const __half*, float*, int, int, hipStream_t);

// hr = (hr - mean_hr) / (std_hr + 1e-8) * std_lr + mean_lr   (fsrcnn_upscaler.py:198-199, :312-313).
// One expression, used by the stand-alone pass and by every consumer that applies it on the fly: the fused and
// the unfused service paths produce bit-identical frames.

__global__ void k_normalize(float* __restrict__ x, const float* __restrict__ st_hr, const float* __restrict__ st_lr,
                            int planes, int hw) {
  const int pl = blockIdx.y;
  const NormCoef k = norm_coef(st_hr, st_lr, pl);
  float* p = x + (size_t)pl * hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)hw; i += (size_t)gridDim.x * blockDim.x)
    p[i] = norm_px(p[i], k);
}
void op_normalize(float* x, const float* st_hr, const float* st_lr, int planes, int hw, hipStream_t st) {
  int gx = std::max(1, std::min((hw + 255) / 256, 1024));
  hipLaunchKernelGGL(k_normalize, dim3(gx, planes), dim3(256), 0, st, x, st_hr, st_lr, planes, hw); SS4K_LAUNCH_OK();
}

// ------------------------------------------------------------------ depthwise KxK, reflect padding
__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

template <int K>
__global__ void k_depthwise_reflect(const float* __restrict__ in, float* __restrict__ out,
                                    const float* __restrict__ taps, int planes, int h, int w, int clamp01,
                                    const float* __restrict__ blend_src, float blend_a, float blend_b) {
  constexpr int r = K >> 1;
  const int y = blockIdx.y, pl = blockIdx.z;
  const float* src = in + (size_t)pl * h * w;
  const size_t row = ((size_t)pl * h + y) * w;
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
    int xi[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) xi[kx] = reflect(x + kx - r, w);
    float acc = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      const float* sr = src + (size_t)reflect(y + ky - r, h) * w;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) acc += taps[ky * K + kx] * sr[xi[kx]];  // same summation order as the flat loop
    }
    if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
    if (blend_src) acc = acc * blend_a + blend_b * blend_src[row + x];
    out[row + x] = acc;
  }
}
void op_depthwise_reflect(const float* in, float* out, const float* taps_dev, int planes, int h, int w, int k,
                          int clamp01, const float* blend_src, float blend_a, float blend_b, hipStream_t st) {
  SS4K_REQUIRE(h <= 65535 && planes <= 65535, "depthwise: grid limits");
  SS4K_REQUIRE(k == 3 || k == 17, "depthwise: kernel size 3 or 17 (the service's sharpen / blur kernels)");
  // torch's reflect padding requires pad < size and raises otherwise (fsrcnn_upscaler.py:20-84 kernels)
  SS4K_REQUIRE(h > k / 2 && w > k / 2, "depthwise reflect: padding (k/2) must be smaller than the plane, as torch requires");
  if (k == 3) {
    hipLaunchKernelGGL(k_depthwise_reflect<3>, grid_rows(w, h, planes), dim3(256), 0, st, in, out, taps_dev, planes, h, w,
                       clamp01, blend_src, blend_a, blend_b);
  } else {
    hipLaunchKernelGGL(k_depthwise_reflect<17>, grid_rows(w, h, planes), dim3(256), 0, st, in, out, taps_dev, planes, h, w,
                       clamp01, blend_src, blend_a, blend_b);
  }
  SS4K_LAUNCH_OK();
}

// 17-tap Gaussian of the local colour match (fsrcnn_upscaler.py:20-52, :211-213) as two 1-D passes: the reference's normalised
// 17x17 kernel is the outer product of g = e / sum(e), e_i = exp(-(i - 8)^2 / (2 sigma^2)), and reflect padding commutes with
// the split.  289 taps -> 34 (the 2-D form stays as the granular op ss4k_op_depthwise_reflect, which takes any taps).
template <bool VERT>
__global__ void k_gauss17(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ g, int planes,
                          int h, int w) {
  const int y = blockIdx.y, pl = blockIdx.z;
  const float* src = in + (size_t)pl * h * w;
  float* dst = out + ((size_t)pl * h + y) * w;
  float gt[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) gt[k] = g[k];
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 17; ++k)
      acc += gt[k] * (VERT ? src[(size_t)reflect(y + k - 8, h) * w + x] : src[(size_t)y * w + reflect(x + k - 8, w)]);
    dst[x] = acc;
  }
}
void op_gauss17_reflect(const float* in, float* tmp, float* out, const float* g17_dev, int planes, int h, int w, hipStream_t st) {
  SS4K_REQUIRE(h <= 65535 && planes <= 65535, "gauss17: grid limits");
  SS4K_REQUIRE(h > 8 && w > 8, "gauss17 reflect: padding (8) must be smaller than the plane, as torch requires");
  hipLaunchKernelGGL(k_gauss17<false>, grid_rows(w, h, planes), dim3(256), 0, st, in, tmp, g17_dev, planes, h, w);
  hipLaunchKernelGGL(k_gauss17<true>, grid_rows(w, h, planes), dim3(256), 0, st, tmp, out, g17_dev, planes, h, w);
  SS4K_LAUNCH_OK();
}

// ------------------------------------------------------------------ bilinear / bicubic (align_corners=False)
// V consecutive outputs per thread (V = 4: 16-byte read-modify-write of the output row when ow % 4 == 0)
template <int V>
__global__ void k_bilinear(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int oh,
                           int ow, int subtract_from_out, int clamp01) {
  const int oy = blockIdx.y, pl = blockIdx.z;
  const float sy = (float)h / oh, sx = (float)w / ow;
  const float* src = in + (size_t)pl * h * w;
  float* dst = out + ((size_t)pl * oh + oy) * ow;
  float fy = sy * (oy + 0.5f) - 0.5f; if (fy < 0) fy = 0;
  const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
  const float ly = fy - y0, hy = 1.f - ly;
  const float* r0 = src + (size_t)y0 * w; const float* r1 = src + (size_t)y1 * w;
  for (int ob = (blockIdx.x * blockDim.x + threadIdx.x) * V; ob < ow; ob += gridDim.x * blockDim.x * V) {
    float cur[V];
    if (subtract_from_out) {
      if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(dst + ob); cur[0] = t.x; cur[1] = t.y; cur[2] = t.z; cur[3] = t.w; }
      else cur[0] = dst[ob];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int ox = ob + e;
      float fx = sx * (ox + 0.5f) - 0.5f; if (fx < 0) fx = 0;
      const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx = fx - x0, hx = 1.f - lx;
      const float v = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
      float r = subtract_from_out ? cur[e] - v : v;
      if (clamp01) r = fminf(fmaxf(r, 0.f), 1.f);
      cur[e] = r;
    }
    if constexpr (V == 4) *reinterpret_cast<float4*>(dst + ob) = make_float4(cur[0], cur[1], cur[2], cur[3]);
    else dst[ob] = cur[0];
  }
}
void op_bilinear(const float* in, float* out, int planes, int h, int w, int oh, int ow, int subtract_from_out,
                 int clamp01, hipStream_t st) {
  SS4K_REQUIRE(oh <= 65535 && planes <= 65535, "bilinear: grid limits");
  if (ow % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(k_bilinear<4>, grid_rows(ow / 4, oh, planes), dim3(256), 0, st, in, out, planes, h, w, oh, ow,
                       subtract_from_out, clamp01);
  } else {
    hipLaunchKernelGGL(k_bilinear<1>, grid_rows(ow, oh, planes), dim3(256), 0, st, in, out, planes, h, w, oh, ow,
                       subtract_from_out, clamp01);
  }
  SS4K_LAUNCH_OK();
}

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
  const float A = -0.75f;
  c[0] = cc2(t + 1.f, A); c[1] = cc1(t, A); c[2] = cc1(1.f - t, A); c[3] = cc2(2.f - t, A);
}
__global__ void k_bicubic(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, int oh,
                          int ow, int clamp01) {
  const int oy = blockIdx.y, pl = blockIdx.z;
  const float sy = (float)h / oh, sx = (float)w / ow;
  const float* src = in + (size_t)pl * h * w;
  float* dst = out + ((size_t)pl * oh + oy) * ow;
  const float fy = sy * (oy + 0.5f) - 0.5f, fly = floorf(fy);
  const int iy = (int)fly;
  float cy[4];
  cubic_coeffs(fy - fly, cy);
  const float* rows[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) rows[a] = src + (size_t)min(max(iy - 1 + a, 0), h - 1) * w;
  for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < ow; ox += gridDim.x * blockDim.x) {
    const float fx = sx * (ox + 0.5f) - 0.5f, flx = floorf(fx);
    const int ix = (int)flx;
    float cx[4]; int xi[4];
    cubic_coeffs(fx - flx, cx);
#pragma unroll
    for (int b = 0; b < 4; ++b) xi[b] = min(max(ix - 1 + b, 0), w - 1);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) row += cx[b] * rows[a][xi[b]];
      acc += cy[a] * row;
    }
    if (clamp01) acc = fminf(fmaxf(acc, 0.f), 1.f);
    dst[ox] = acc;
  }
}
void op_bicubic(const float* in, float* out, int planes, int h, int w, int oh, int ow, int clamp01, hipStream_t st) {
  SS4K_REQUIRE(oh <= 65535 && planes <= 65535, "bicubic: grid limits");
  hipLaunchKernelGGL(k_bicubic, grid_rows(ow, oh, planes), dim3(256), 0, st, in, out, planes, h, w, oh, ow, clamp01); SS4K_LAUNCH_OK();
}

// ------------------------------------------------------------------ fused tails of the service path
// One pass from the network's raw fp32 output to the next tensor of the path, applying in the reference's order
//   [normalise (:198-199)] -> [- bilinear(diff) (:214-217)] -> clamp(0,1) (:220) -> [* 255, truncate to uint8 NHWC (:232-233)]
// with exactly the per-element expressions of the stand-alone kernels (k_normalize, k_bilinear, k_clamp01,
// k_f32nchw_to_u8nhwc), so the frames are bit-identical to the unfused path.
// U8: write uint8 NHWC (three planes of a pixel by one thread); else write the clamped float back in place.
template <bool NORM, bool DIFF, bool U8, typename HT = float>
__global__ void k_tail_fused(HT* __restrict__ hr, uint8_t* __restrict__ out, const float* __restrict__ diff, int n, int c, int h, int w,
                             int dh, int dw, const float* __restrict__ st_hr, const float* __restrict__ st_lr) {
  const int oy = blockIdx.y, img = blockIdx.z;
  // bilinear, align_corners=False: same arithmetic as k_bilinear
  const float sy = (float)dh / h, sx = (float)dw / w;
  float fy = sy * (oy + 0.5f) - 0.5f; if (fy < 0) fy = 0;
  const int y0 = (int)fy, y1 = y0 + (y0 < dh - 1 ? 1 : 0);
  const float ly = fy - y0, hy = 1.f - ly;
  for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < w; ox += gridDim.x * blockDim.x) {
    float fx = sx * (ox + 0.5f) - 0.5f; if (fx < 0) fx = 0;
    const int x0 = (int)fx, x1 = x0 + (x0 < dw - 1 ? 1 : 0);
    const float lx = fx - x0, hx = 1.f - lx;
    for (int k = 0; k < c; ++k) {
      const int pl = img * c + k;
      const size_t idx = ((size_t)pl * h + oy) * w + ox;
      float v = hr_ld(hr + idx);
      if constexpr (NORM) v = norm_px(v, norm_coef(st_hr, st_lr, pl));
      if constexpr (DIFF) {
        const float* r0 = diff + ((size_t)pl * dh + y0) * dw; const float* r1 = diff + ((size_t)pl * dh + y1) * dw;
        const float d = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        v = v - d;
      }
      v = fminf(fmaxf(v, 0.f), 1.f);
      if constexpr (U8) out[(((size_t)img * h + oy) * w + ox) * c + k] = (uint8_t)(fminf(fmaxf(v, 0.f), 1.f) * 255.f);
      else hr_st(hr + idx, v);
    }
  }
}
// the same pass with four consecutive pixels per thread (w % 4 == 0): 16-byte loads of each plane, one 12-byte store of
// the four uint8 NHWC pixels (or a 16-byte store per plane); per-element expressions and their order are k_tail_fused's
template <bool NORM, bool DIFF, bool U8, typename HT = float>
__global__ void k_tail_fused4(HT* __restrict__ hr, uint8_t* __restrict__ out, const float* __restrict__ diff, int n, int h, int w,
                              int dh, int dw, const float* __restrict__ st_hr, const float* __restrict__ st_lr) {
  constexpr int C = 3;
  const int oy = blockIdx.y, img = blockIdx.z;
  const float sy = (float)dh / h, sx = (float)dw / w;
  float fy = sy * (oy + 0.5f) - 0.5f; if (fy < 0) fy = 0;
  const int y0 = (int)fy, y1 = y0 + (y0 < dh - 1 ? 1 : 0);
  const float ly = fy - y0, hy = 1.f - ly;
  for (int ox4 = blockIdx.x * blockDim.x + threadIdx.x; ox4 < w / 4; ox4 += gridDim.x * blockDim.x) {
    int x0[4], x1[4]; float lx[4], hx[4];
    if constexpr (DIFF) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float fx = sx * (4 * ox4 + j + 0.5f) - 0.5f; if (fx < 0) fx = 0;
        x0[j] = (int)fx; x1[j] = x0[j] + (x0[j] < dw - 1 ? 1 : 0);
        lx[j] = fx - x0[j]; hx[j] = 1.f - lx[j];
      }
    }
    uint32_t b[C][4];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const int pl = img * C + k;
      HT* p = hr + ((size_t)pl * h + oy) * w + 4 * (size_t)ox4;
      const float4 in4 = hr_ld4<HT>(p);
      float v[4] = {in4.x, in4.y, in4.z, in4.w};
      NormCoef nk{};
      if constexpr (NORM) nk = norm_coef(st_hr, st_lr, pl);
      const float* r0 = nullptr; const float* r1 = nullptr;
      if constexpr (DIFF) { r0 = diff + ((size_t)pl * dh + y0) * dw; r1 = diff + ((size_t)pl * dh + y1) * dw; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (NORM) v[j] = norm_px(v[j], nk);
        if constexpr (DIFF) {
          const float d = hy * (hx[j] * r0[x0[j]] + lx[j] * r0[x1[j]]) + ly * (hx[j] * r1[x0[j]] + lx[j] * r1[x1[j]]);
          v[j] = v[j] - d;
        }
        v[j] = fminf(fmaxf(v[j], 0.f), 1.f);
        if constexpr (U8) b[k][j] = (uint32_t)(uint8_t)(fminf(fmaxf(v[j], 0.f), 1.f) * 255.f);
      }
      if constexpr (!U8) hr_st4<HT>(p, make_float4(v[0], v[1], v[2], v[3]));
    }
    if constexpr (U8) {
      // bytes 3*j + k of the twelve: pixel j, plane k
      uint32_t wds[3] = {0u, 0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < C; ++k) { const int byte = 3 * j + k; wds[byte >> 2] |= b[k][j] << (8 * (byte & 3)); }
      uint32_t* o = reinterpret_cast<uint32_t*>(out + (((size_t)img * h + oy) * w + 4 * (size_t)ox4) * C);
      o[0] = wds[0]; o[1] = wds[1]; o[2] = wds[2];
    }
  }
}
template <typename HT>
void op_tail_fused(HT* hr, uint8_t* out_u8, const float* diff, int n, int c, int h, int w, int dh, int dw, const float* st_hr,
                   const float* st_lr, hipStream_t st) {
  SS4K_REQUIRE(h <= 65535 && n <= 65535, "fused tail: grid limits");
  const dim3 g = grid_rows(w, h, n);
  const bool norm = st_hr != nullptr, df = diff != nullptr, u8 = out_u8 != nullptr;
  if (c == 3 && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(hr) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_u8) & 3) == 0) {
    const dim3 g4 = grid_rows(w / 4, h, n);
#define SS4K_TAIL4(N_, D_, U_) hipLaunchKernelGGL((k_tail_fused4<N_, D_, U_, HT>), g4, dim3(256), 0, st, hr, out_u8, diff, n, h, w, dh, dw, st_hr, st_lr)
    if (norm && df && u8) SS4K_TAIL4(true, true, true); else if (norm && df) SS4K_TAIL4(true, true, false);
    else if (norm && u8) SS4K_TAIL4(true, false, true); else if (norm) SS4K_TAIL4(true, false, false);
    else if (df && u8) SS4K_TAIL4(false, true, true); else if (df) SS4K_TAIL4(false, true, false);
    else if (u8) SS4K_TAIL4(false, false, true); else SS4K_TAIL4(false, false, false);
#undef SS4K_TAIL4
    SS4K_LAUNCH_OK();
    return;
  }
#define SS4K_TAIL(N_, D_, U_) hipLaunchKernelGGL((k_tail_fused<N_, D_, U_, HT>), g, dim3(256), 0, st, hr, out_u8, diff, n, c, h, w, dh, dw, st_hr, st_lr)
  if (norm && df && u8) SS4K_TAIL(true, true, true); else if (norm && df) SS4K_TAIL(true, true, false);
  else if (norm && u8) SS4K_TAIL(true, false, true); else if (norm) SS4K_TAIL(true, false, false);
  else if (df && u8) SS4K_TAIL(false, true, true); else if (df) SS4K_TAIL(false, true, false);
  else if (u8) SS4K_TAIL(false, false, true); else SS4K_TAIL(false, false, false);
#undef SS4K_TAIL
  SS4K_LAUNCH_OK();
}
template void op_tail_fused<float>(float*, uint8_t*, const float*, int, int, int, int, int, int, const float*, const float*, hipStream_t);
template void op_tail_fused<__half>(__half*, uint8_t*, const float*, int, int, int, int, int, int, const float*, const float*, hipStream_t);

// bicubic (A = -0.75, align_corners=False: k_bicubic's arithmetic) of an already clamped tensor, then
// clamp(0,1) * 255 truncated to uint8 NHWC: the resized float tensor is never written
template <typename HT = float>
__global__ void k_bicubic_u8(const HT* __restrict__ in, uint8_t* __restrict__ out, int n, int c, int h, int w, int oh, int ow) {
  const int oy = blockIdx.y, img = blockIdx.z;
  const float sy = (float)h / oh, sx = (float)w / ow;
  const float fy = sy * (oy + 0.5f) - 0.5f, fly = floorf(fy);
  const int iy = (int)fly;
  float cy[4];
  cubic_coeffs(fy - fly, cy);
  int yi[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) yi[a] = min(max(iy - 1 + a, 0), h - 1);
  for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < ow; ox += gridDim.x * blockDim.x) {
    const float fx = sx * (ox + 0.5f) - 0.5f, flx = floorf(fx);
    const int ix = (int)flx;
    float cx[4]; int xi[4];
    cubic_coeffs(fx - flx, cx);
#pragma unroll
    for (int b = 0; b < 4; ++b) xi[b] = min(max(ix - 1 + b, 0), w - 1);
    for (int k = 0; k < c; ++k) {
      const HT* src = in + (size_t)(img * c + k) * h * w;
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const HT* rowp = src + (size_t)yi[a] * w;
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) row += cx[b] * hr_ld(rowp + xi[b]);
        acc += cy[a] * row;
      }
      acc = fminf(fmaxf(acc, 0.f), 1.f);
      out[(((size_t)img * oh + oy) * ow + ox) * c + k] = (uint8_t)(fminf(fmaxf(acc, 0.f), 1.f) * 255.f);
    }
  }
}
// the same resize at exactly 2 : 1 (the x4 network's output brought to the x2 frame size, fsrcnn_upscaler.py:223-231):
// every sample sits at t = 0.5 of input columns 2*ox - 1 .. 2*ox + 2, so four adjacent outputs share ten input columns of
// each of their four rows - two 16-byte loads and two clamped edge loads instead of sixteen 4-byte ones, and one 12-byte
// store of the four uint8 NHWC pixels.  Coefficients, products and their order are k_bicubic_u8's.
template <typename HT = float>
__global__ void k_bicubic_u8_half(const HT* __restrict__ in, uint8_t* __restrict__ out, int n, int h, int w, int oh, int ow) {
  constexpr int C = 3;
  const int oy = blockIdx.y, img = blockIdx.z;
  const float sy = (float)h / oh, sx = (float)w / ow;
  const float fy = sy * (oy + 0.5f) - 0.5f, fly = floorf(fy);
  const int iy = (int)fly;
  float cy[4];
  cubic_coeffs(fy - fly, cy);
  int yi[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) yi[a] = min(max(iy - 1 + a, 0), h - 1);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ow / 4; t += gridDim.x * blockDim.x) {
    float cx[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float fx = sx * (4 * t + j + 0.5f) - 0.5f;
      cubic_coeffs(fx - floorf(fx), cx[j]);
    }
    const int xl = max(8 * t - 1, 0), xr = min(8 * t + 8, w - 1);
    uint32_t b[C][4];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const HT* src = in + (size_t)(img * C + k) * h * w;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const HT* rowp = src + (size_t)yi[a] * w;
        const float4 m0 = hr_ld4<HT>(rowp + 8 * t), m1 = hr_ld4<HT>(rowp + 8 * t + 4);
        const float col[10] = {hr_ld(rowp + xl), m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, hr_ld(rowp + xr)};   // columns 8t-1 .. 8t+8
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float row = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) row += cx[j][q] * col[2 * j + q];
          acc[j] += cy[a] * row;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = fminf(fmaxf(acc[j], 0.f), 1.f);
        b[k][j] = (uint32_t)(uint8_t)(fminf(fmaxf(v, 0.f), 1.f) * 255.f);
      }
    }
    uint32_t wds[3] = {0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < C; ++k) { const int byte = 3 * j + k; wds[byte >> 2] |= b[k][j] << (8 * (byte & 3)); }
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (((size_t)img * oh + oy) * ow + 4 * (size_t)t) * C);
    o[0] = wds[0]; o[1] = wds[1]; o[2] = wds[2];
  }
}
template <typename HT>
void op_bicubic_u8(const HT* in, uint8_t* out, int n, int c, int h, int w, int oh, int ow, hipStream_t st) {
  SS4K_REQUIRE(oh <= 65535 && n <= 65535, "bicubic: grid limits");
  if (c == 3 && h == 2 * oh && w == 2 * ow && (ow & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
    hipLaunchKernelGGL(k_bicubic_u8_half<HT>, grid_rows(ow / 4, oh, n), dim3(256), 0, st, in, out, n, h, w, oh, ow); SS4K_LAUNCH_OK();
    return;
  }
  hipLaunchKernelGGL(k_bicubic_u8<HT>, grid_rows(ow, oh, n), dim3(256), 0, st, in, out, n, c, h, w, oh, ow); SS4K_LAUNCH_OK();
}
template void op_bicubic_u8<float>(const float*, uint8_t*, int, int, int, int, int, int, hipStream_t);
template void op_bicubic_u8<__half>(const __half*, uint8_t*, int, int, int, int, int, int, hipStream_t);

// ------------------------------------------------------------------ elementwise helpers
__global__ void k_sub(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = a[i] - b[i];
}
void op_sub(const float* a, const float* b, float* out, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_sub, grid1d(n), dim3(256), 0, st, a, b, out, n); SS4K_LAUNCH_OK();
}
__global__ void k_clamp01(float* __restrict__ x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}
void op_clamp01(float* x, size_t n, hipStream_t st) { hipLaunchKernelGGL(k_clamp01, grid1d(n), dim3(256), 0, st, x, n); SS4K_LAUNCH_OK(); }

// cv2.resize(img, None, fx, fy, INTER_AREA) on uint8 NHWC frames, shrinking by a non-integer factor: the image server's pre / post scale
// (image_pipeline.py:272-273, 347-348).  OpenCV's general area path (modules/imgproc/src/resize.cpp, computeResizeAreaTab +
// ResizeArea_Invoker<uchar, float>; restated on the CPU in oracle/cv_area.py - PARITY UNPINNED: cv2 is not in the image): per source row
// buf = sum of S * alpha over the cell's x entries in table order, per output row sum = beta * buf for the first y entry, += for the
// others, float32, multiply and add rounded separately (OpenCV's baseline build has no FMA), result rounded half to even and clipped.
// ent: {source index, weight} per table entry; ofs[d] .. ofs[d + 1]: the entries of output index d.
struct CvAreaEnt { int si; float a; };
__global__ void k_cv_area_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const CvAreaEnt* __restrict__ xe, const int* __restrict__ xo,
                             const CvAreaEnt* __restrict__ ye, const int* __restrict__ yo, int n, int h, int w, int c, int oh, int ow) {
#pragma clang fp contract(off)   // multiply and add are rounded separately, as in OpenCV's baseline build (plain operators under this pragma: HIP's
                                 // __fmul_rn / __fadd_rn are inline functions compiled under the header's own contraction mode and DO get fused)
  const size_t total = (size_t)n * oh * ow;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int dx = (int)(i % ow), dy = (int)((i / ow) % oh), img = (int)(i / ((size_t)ow * oh));
    const uint8_t* src = in + (size_t)img * h * w * c;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    const int x0 = xo[dx], x1 = xo[dx + 1], y0 = yo[dy], y1 = yo[dy + 1];
    for (int j = y0; j < y1; ++j) {
      const uint8_t* row = src + (size_t)ye[j].si * w * c;
      const float beta = ye[j].a;
      float buf[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = x0; k < x1; ++k) {
        const uint8_t* px = row + (size_t)xe[k].si * c;
        const float alpha = xe[k].a;
        for (int ch = 0; ch < c; ++ch) { const float p = (float)px[ch] * alpha; buf[ch] = buf[ch] + p; }
      }
      for (int ch = 0; ch < c; ++ch) { const float p = beta * buf[ch]; sum[ch] = j == y0 ? p : sum[ch] + p; }
    }
    for (int ch = 0; ch < c; ++ch) out[i * c + ch] = (uint8_t)fminf(fmaxf(rintf(sum[ch]), 0.f), 255.f);
  }
}
void op_cv_area_u8(const uint8_t* in, uint8_t* out, const void* xe, const int* xo, const void* ye, const int* yo, int n, int h, int w, int c, int oh, int ow,
                   hipStream_t st) {
  hipLaunchKernelGGL(k_cv_area_u8, grid1d((size_t)n * oh * ow), dim3(256), 0, st, in, out, (const CvAreaEnt*)xe, xo, (const CvAreaEnt*)ye, yo, n, h, w, c, oh, ow);
  SS4K_LAUNCH_OK();
}

// one wave that occupies its stream for `ticks` of the 100 MHz real-time counter and does nothing else (ss4k_ctx::lane_check times a
// pair of these to see whether two streams run side by side); bounded: every wave leaves after at most LANE_SPIN_MAX_TICKS
constexpr unsigned LANE_SPIN_MAX_TICKS = 50000;   // 0.5 ms
__global__ void k_lane_spin(unsigned ticks) {
  ticks = ticks < LANE_SPIN_MAX_TICKS ? ticks : LANE_SPIN_MAX_TICKS;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
void op_lane_spin(unsigned ticks, hipStream_t st) { hipLaunchKernelGGL(k_lane_spin, dim3(1), dim3(64), 0, st, ticks); SS4K_LAUNCH_OK(); }

// (clamp(x,0,1)*255) -> uint8 by truncation, NCHW -> NHWC   (fsrcnn_upscaler.py:232-233, :325-326)
__global__ void k_f32nchw_to_u8nhwc(const float* __restrict__ in, uint8_t* __restrict__ out, int n, int c, int h,
                                    int w) {
  const size_t hw = (size_t)h * w, total = (size_t)n * hw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / hw, p = i - img * hw;
    for (int k = 0; k < c; ++k) {
      float v = in[(img * c + k) * hw + p];
      v = fminf(fmaxf(v, 0.f), 1.f) * 255.f;
      out[i * c + k] = (uint8_t)v;
    }
  }
}
void op_f32nchw_to_u8nhwc(const float* in, uint8_t* out, int n, int c, int h, int w, hipStream_t st) {
  hipLaunchKernelGGL(k_f32nchw_to_u8nhwc, grid1d((size_t)n * h * w), dim3(256), 0, st, in, out, n, c, h, w); SS4K_LAUNCH_OK();
}

// ------------------------------------------------------------------ network input / output layout converters
// NCHW fp32 -> "planes" T (conv_mfma.hip): channel k lives in plane k / CW at offset k % CW of the
// pixel's 64-byte record; optional pixel-unshuffle(r) (RRDBNet x2/x1 front end, basicsr
// pixel_unshuffle: channel = c*r*r + dy*r + dx); unused channels of the last plane are zeroed.
template <typename T, int R>
__global__ void k_pack_input(const float* __restrict__ in, T* __restrict__ out, int n, int c, int h, int w,
                             int nplanes) {
  constexpr int CW = 16, RV = CW * sizeof(T) / 16;  // 16-channel records of RV 16-byte slots
  const int oh = h / R, ow = w / R;
  const size_t total = (size_t)n * oh * ow;
  const int creal = c * R * R;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = i % ow, oy = (i / ow) % oh;
    const size_t img = i / ((size_t)ow * oh);
    for (int p = 0; p < nplanes; ++p) {
      T rec[CW];  // one record, written with 16-byte stores
#pragma unroll
      for (int q = 0; q < CW; ++q) {
        const int k = p * CW + q;
        float v = 0.f;
        if (k < creal) {
          const int ci = k / (R * R), rem = k % (R * R), dy = rem / R, dx = rem % R;
          v = in[((img * c + ci) * h + (size_t)oy * R + dy) * w + (size_t)ox * R + dx];
        }
        rec[q] = (T)v;
      }
      uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)p * total + i) * CW);
      const uint4* src = reinterpret_cast<const uint4*>(rec);
#pragma unroll
      for (int q = 0; q < RV; ++q) dst[q] = src[q];
    }
  }
}
template <typename T>
void op_pack_input(const float* in, T* out, int n, int c, int h, int w, int r, int nplanes, hipStream_t st) {
  const dim3 g = grid1d((size_t)n * (h / r) * (w / r));
  if (r == 1) { hipLaunchKernelGGL((k_pack_input<T, 1>), g, dim3(256), 0, st, in, out, n, c, h, w, nplanes); }
  else if (r == 2) { hipLaunchKernelGGL((k_pack_input<T, 2>), g, dim3(256), 0, st, in, out, n, c, h, w, nplanes); }
  else if (r == 4) { hipLaunchKernelGGL((k_pack_input<T, 4>), g, dim3(256), 0, st, in, out, n, c, h, w, nplanes); }
  else throw Error(SS4K_EINVAL, "pack_input: unshuffle factor must be 1, 2 or 4");
  SS4K_LAUNCH_OK();
}
template void op_pack_input<float>(const float*, float*, int, int, int, int, int, int, hipStream_t);
template void op_pack_input<__half>(const float*, __half*, int, int, int, int, int, int, hipStream_t);

// SRVGGNetCompact tail (realesrgan/factory.py:77-81): PixelShuffle(r) of a "planes" T tensor into
// fp32 NCHW planes plus the nearest-upsampled network input.
// One thread per LR pixel and colour: the r*r channels of that colour are one contiguous run of its
// record(s), read with 16-byte loads, and leave as r rows of r floats (16-byte stores for r = 4).
template <typename T, int R, bool STATS, typename HT = float>
__global__ __launch_bounds__(256) void k_ps_nchw_addbase(const T* __restrict__ src, HT* __restrict__ out, const float* __restrict__ base,
                                  int n, int h, int w, int cq, double* __restrict__ acc) {
  constexpr int CW = 16, RR = R * R;
  const int y = blockIdx.y, img = blockIdx.z;
  const int OH = h * R, OW = w * R;
  const size_t npix = (size_t)n * h * w;
  double ps[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};   // per colour (cq <= 4) partial sum / sum of squares of this thread
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
    const size_t pix = ((size_t)img * h + y) * w + x;
    for (int c = 0; c < cq; ++c) {
      const int ch0 = c * RR;  // RR consecutive channels, never straddling a 16-channel record (RR | 16)
      const T* rec = src + ((size_t)(ch0 / CW) * npix + pix) * CW + (ch0 % CW);
      T v[RR];
      if constexpr (RR * sizeof(T) >= 16) {
#pragma unroll
        for (int q = 0; q < (int)(RR * sizeof(T) / 16); ++q) reinterpret_cast<uint4*>(v)[q] = reinterpret_cast<const uint4*>(rec)[q];
      } else {
        *reinterpret_cast<uint2*>(v) = *reinterpret_cast<const uint2*>(rec);  // 4 fp16 channels
      }
      const float b = base[(((size_t)img * cq + c) * h + y) * w + x];
      HT* o = out + (((size_t)img * cq + c) * OH + (size_t)y * R) * OW + (size_t)x * R;
      float f[RR];
#pragma unroll
      for (int k = 0; k < RR; ++k) f[k] = (float)v[k] + b;
#pragma unroll
      for (int dy = 0; dy < R; ++dy) {
        if constexpr (R == 4) hr_st4<HT>(o + (size_t)dy * OW, make_float4(f[dy * 4], f[dy * 4 + 1], f[dy * 4 + 2], f[dy * 4 + 3]));
        else { hr_st(o + (size_t)dy * OW, f[dy * 2]); hr_st(o + (size_t)dy * OW + 1, f[dy * 2 + 1]); }
      }
      if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < RR; ++k) { const double d = f[k]; ps[c] += d; pq[c] += d * d; }
      }
    }
  }
  if constexpr (STATS) {
    // plane statistics of the tensor being written (fsrcnn_upscaler.py:192-197) ride along: fp64 partials per
    // workgroup, one atomic pair per colour - the separate read pass over the HR tensor is gone
    __shared__ double ss[4][4], sq[4][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = 0; c < cq; ++c) {
      double s_ = ps[c], q_ = pq[c];
      for (int off = 32; off > 0; off >>= 1) { s_ += __shfl_down(s_, off, 64); q_ += __shfl_down(q_, off, 64); }
      if (lane == 0) { ss[c][wv] = s_; sq[c][wv] = q_; }
    }
    __syncthreads();
    if (threadIdx.x < cq) {
      const int c = threadIdx.x;
      double S = 0, Q = 0;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { S += ss[c][i]; Q += sq[c][i]; }
      double* a = acc + ((size_t)((blockIdx.x + blockIdx.y) % STATS_SLOTS) * (gridDim.z * cq) + img * cq + c) * 2;
      atomicAdd(&a[0], S);
      atomicAdd(&a[1], Q);
    }
  }
}
template <typename T, typename HT>
void op_ps_nchw_addbase(const T* src, HT* out, const float* base, int n, int h, int w, int r, int cq, double* stats_acc, hipStream_t st) {
  SS4K_REQUIRE(h <= 65535 && n <= 65535, "pixel shuffle tail: grid limits");
  SS4K_REQUIRE(!stats_acc || cq <= 4, "pixel shuffle tail: statistics for at most 4 colours");
  if (stats_acc) SS4K_HIP(hipMemsetAsync(stats_acc, 0, sizeof(double) * 2 * n * cq * STATS_SLOTS, st));
  if (r == 4) {
    if (stats_acc) { hipLaunchKernelGGL((k_ps_nchw_addbase<T, 4, true, HT>), grid_rows(w, h, n), dim3(256), 0, st, src, out, base, n, h, w, cq, stats_acc); }
    else { hipLaunchKernelGGL((k_ps_nchw_addbase<T, 4, false, HT>), grid_rows(w, h, n), dim3(256), 0, st, src, out, base, n, h, w, cq, stats_acc); }
  } else if (r == 2) {
    if (stats_acc) { hipLaunchKernelGGL((k_ps_nchw_addbase<T, 2, true, HT>), grid_rows(w, h, n), dim3(256), 0, st, src, out, base, n, h, w, cq, stats_acc); }
    else { hipLaunchKernelGGL((k_ps_nchw_addbase<T, 2, false, HT>), grid_rows(w, h, n), dim3(256), 0, st, src, out, base, n, h, w, cq, stats_acc); }
  } else throw Error(SS4K_EINVAL, "SRVGG: upscale must be 2 or 4");
  SS4K_LAUNCH_OK();
}
template void op_ps_nchw_addbase<float, float>(const float*, float*, const float*, int, int, int, int, int, double*, hipStream_t);
template void op_ps_nchw_addbase<__half, float>(const __half*, float*, const float*, int, int, int, int, int, double*, hipStream_t);
template void op_ps_nchw_addbase<__half, __half>(const __half*, __half*, const float*, int, int, int, int, int, double*, hipStream_t);

// BSVD stream mode: the ShiftConv input of frame t (bsvd/model.py:42-53,95-138) takes channels
// [0, fold) from frame t+1, [fold, 2*fold) from frame t-1 (zeros past either end of the stream) and
// the rest from frame t.  Only the leading planes that hold channels < 2*fold are rebuilt here (the
// conv reads the remaining planes from the tensor itself); one thread per 16-byte slot.
__global__ void k_temporal_shift(const uint4* __restrict__ in, uint4* __restrict__ out, int nplanes, int frames,
                                 size_t frame_px, int spr, int ch_per_plane, int fold) {
  const int ch_per_slot = ch_per_plane / spr;
  const size_t slots_per_plane = (size_t)frames * frame_px * spr;
  const size_t total = slots_per_plane * nplanes;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(i / slots_per_plane);
    const size_t r = i - (size_t)p * slots_per_plane;
    const int t = (int)(r / (frame_px * spr));
    const int ch0 = p * ch_per_plane + (int)(r % spr) * ch_per_slot;
    const int ts = ch0 < fold ? t + 1 : (ch0 < 2 * fold ? t - 1 : t);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ts >= 0 && ts < frames) v = in[i + (ptrdiff_t)(ts - t) * (ptrdiff_t)(frame_px * spr)];
    out[i] = v;
  }
}
void op_temporal_shift(const void* in, void* out, int nplanes, int frames, size_t frame_px, int slots_per_record,
                       int ch_per_plane, int fold, hipStream_t st) {
  if (fold % (ch_per_plane / slots_per_record) != 0)
    throw Error(SS4K_EINVAL, "temporal shift: fold must be a multiple of the 16-byte channel group");
  hipLaunchKernelGGL(k_temporal_shift, grid1d((size_t)nplanes * frames * frame_px * slots_per_record), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(in), reinterpret_cast<uint4*>(out), nplanes, frames, frame_px,
                     slots_per_record, ch_per_plane, fold); SS4K_LAUNCH_OK();
}

}  // namespace ss4k
