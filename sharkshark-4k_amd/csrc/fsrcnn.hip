// FSRCNN forward (reference src/upscale/model/fsrcnn/model.py:14-62) on fp32 planes.
// Channel depths are 1 / 56 / 12, far too shallow for a dense MFMA contraction, so these are
// direct convolutions on the vector ALUs: one thread per low-resolution pixel holds every output
// channel in registers, weights are wave-uniform (scalar loads), activations are stored as groups of
// four channels, group-major ([C/4][pixel][4] fp32), so a wave's 16-byte loads/stores of one group are
// one contiguous kilobyte whatever the channel count.  Layers are fused where no halo is needed:
//   head  = conv5x5(1->56)+PReLU -> conv1x1(56->12)+PReLU        (56-wide map never leaves registers)
//   map   = conv3x3(12->12)+PReLU                                 (x4)
//   tail  = conv1x1(12->56)+PReLU                                 (NHWC 56 for the deconv gather)
//   deconv= ConvTranspose 9x9 stride s as s*s sub-pixel phases, all phases of one LR pixel per thread
#include "common.h"
#include "glue.h"

namespace ss4k {

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }

__global__ __launch_bounds__(256) void k_fs_head(const float* __restrict__ in, float* __restrict__ out,
                                                 const float* __restrict__ wf, const float* __restrict__ bf,
                                                 const float* __restrict__ af, const float* __restrict__ ws,
                                                 const float* __restrict__ bs, const float* __restrict__ as,
                                                 int planes, int h, int w) {
  const size_t total = (size_t)planes * h * w;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % w, y = (i / w) % h;
  const float* src = in + (i / ((size_t)w * h)) * (size_t)h * w;
  float f[56];
#pragma unroll
  for (int c = 0; c < 56; ++c) f[c] = bf[c];
  // A real loop over the 25 taps (not unrolled): one tap's 56 scalar weights are live at a time.
  // Fully unrolled, hipcc hoists all 1400 invariant scalar loads to the top of the kernel and spills
  // the SGPRs into VGPR lanes (2 x 1340 v_writelane/v_readlane per thread, 4x the useful work).
#pragma unroll 1
  for (int t = 0; t < 25; ++t) {
    const int ky = t / 5, kx = t - ky * 5;
    const int yy = y + ky - 2, xx = x + kx - 2;
    const float v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? src[(size_t)yy * w + xx] : 0.f;
    const float* wt = wf + t * 56;
#pragma unroll
    for (int c = 0; c < 56; ++c) f[c] = fmaf(wt[c], v, f[c]);
  }
#pragma unroll
  for (int c = 0; c < 56; ++c) f[c] = prelu(f[c], af[c]);
  float s[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) s[j] = bs[j];
#pragma unroll
  for (int c = 0; c < 56; ++c)
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = fmaf(ws[c * 12 + j], f[c], s[j]);
  float4* dst = reinterpret_cast<float4*>(out);
#pragma unroll
  for (int q = 0; q < 3; ++q)
    dst[q * total + i] = make_float4(prelu(s[4 * q], as[4 * q]), prelu(s[4 * q + 1], as[4 * q + 1]),
                                     prelu(s[4 * q + 2], as[4 * q + 2]), prelu(s[4 * q + 3], as[4 * q + 3]));
}

__global__ __launch_bounds__(256) void k_fs_map(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ wm, const float* __restrict__ bm,
                                                const float* __restrict__ am, int planes, int h, int w) {
  const size_t total = (size_t)planes * h * w;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % w, y = (i / w) % h;
  const size_t pbase = (i / ((size_t)w * h)) * (size_t)h * w;
  float s[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) s[j] = bm[j];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = y + ky - 1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = x + kx - 1;
      if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      const float4* p = reinterpret_cast<const float4*>(in) + pbase + (size_t)yy * w + xx;
      float v[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) { const float4 t = p[q * total]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
#pragma unroll
      for (int c = 0; c < 12; ++c)
#pragma unroll
        for (int j = 0; j < 12; ++j) s[j] = fmaf(wm[((ky * 3 + kx) * 12 + c) * 12 + j], v[c], s[j]);
    }
  }
  float4* dst = reinterpret_cast<float4*>(out) + i;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    dst[q * total] = make_float4(prelu(s[4 * q], am[4 * q]), prelu(s[4 * q + 1], am[4 * q + 1]),
                         prelu(s[4 * q + 2], am[4 * q + 2]), prelu(s[4 * q + 3], am[4 * q + 3]));
}

__global__ __launch_bounds__(256) void k_fs_expand(const float* __restrict__ in, float* __restrict__ out,
                                                   const float* __restrict__ we, const float* __restrict__ be,
                                                   const float* __restrict__ ae, size_t total) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float4* p = reinterpret_cast<const float4*>(in) + i;
  float v[12];
#pragma unroll
  for (int q = 0; q < 3; ++q) { const float4 t = p[q * total]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
  float4* dst = reinterpret_cast<float4*>(out) + i;
#pragma unroll
  for (int q = 0; q < 14; ++q) {
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      float s = be[c];
#pragma unroll
      for (int k = 0; k < 12; ++k) s = fmaf(we[k * 56 + c], v[k], s);
      o[e] = prelu(s, ae[c]);
    }
    dst[q * total] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

template <int S>
__global__ __launch_bounds__(256) void k_fs_deconv(const float* __restrict__ in, float* __restrict__ out,
                                                   const float* __restrict__ wd, float bias, int planes, int h,
                                                   int w) {
  // the 81x56 taps (18 KB) overflow the 16 KB scalar cache when read as wave-uniform scalars (every
  // s_load then misses to L2): keep them in LDS and read them as broadcasts instead
  __shared__ float4 w_lds[81 * 14];
  for (int k = threadIdx.x; k < 81 * 14; k += blockDim.x) w_lds[k] = reinterpret_cast<const float4*>(wd)[k];
  __syncthreads();
  const size_t total = (size_t)planes * h * w;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % w, y = (i / w) % h;
  const size_t pl = i / ((size_t)w * h);
  const size_t pbase = pl * (size_t)h * w;
  float acc[S][S];
#pragma unroll
  for (int a = 0; a < S; ++a)
#pragma unroll
    for (int b = 0; b < S; ++b) acc[a][b] = bias;
  // out[s*y+py] gathers in[y+d] through tap ky = py + 4 - s*d  (stride s, padding 4, kernel 9)
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= h) continue;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= w) continue;
      const float4* p = reinterpret_cast<const float4*>(in) + pbase + (size_t)yy * w + xx;
      float4 v[14];
#pragma unroll
      for (int q = 0; q < 14; ++q) v[q] = p[q * total];
#pragma unroll
      for (int py = 0; py < S; ++py) {
        const int ky = py + 4 - S * dy;
        if (ky < 0 || ky > 8) continue;
#pragma unroll
        for (int px = 0; px < S; ++px) {
          const int kx = px + 4 - S * dx;
          if (kx < 0 || kx > 8) continue;
          const float4* wv = w_lds + (ky * 9 + kx) * 14;
          float s = acc[py][px];
#pragma unroll
          for (int q = 0; q < 14; ++q) {
            const float4 t = wv[q];
            s = fmaf(t.x, v[q].x, s); s = fmaf(t.y, v[q].y, s); s = fmaf(t.z, v[q].z, s); s = fmaf(t.w, v[q].w, s);
          }
          acc[py][px] = s;
        }
      }
    }
  }
  const size_t OW = (size_t)w * S;
  float* dst = out + pl * (size_t)h * S * OW + (size_t)y * S * OW + (size_t)x * S;
#pragma unroll
  for (int py = 0; py < S; ++py)
#pragma unroll
    for (int px = 0; px < S; ++px) dst[py * OW + px] = acc[py][px];
}

void fsrcnn_forward(ss4k_ctx*, const FsrcnnWeights& W, int factor, const float* in, float* out, int planes, int h,
                    int w, float* ws12a, float* ws12b, float* ws56, hipStream_t st) {
  const size_t total = (size_t)planes * h * w;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_fs_head, grid, block, 0, st, in, ws12a, W.w_feat, W.b_feat, W.a_feat, W.w_shrink, W.b_shrink,
                     W.a_shrink, planes, h, w);
  float* cur = ws12a; float* nxt = ws12b;
  for (int l = 0; l < 4; ++l) {
    hipLaunchKernelGGL(k_fs_map, grid, block, 0, st, cur, nxt, W.w_map[l], W.b_map[l], W.a_map[l], planes, h, w);
    std::swap(cur, nxt);
  }
  hipLaunchKernelGGL(k_fs_expand, grid, block, 0, st, cur, ws56, W.w_expand, W.b_expand, W.a_expand, total);
  if (factor == 2)
    hipLaunchKernelGGL((k_fs_deconv<2>), grid, block, 0, st, ws56, out, W.w_deconv, W.b_deconv, planes, h, w);
  else if (factor == 4)
    hipLaunchKernelGGL((k_fs_deconv<4>), grid, block, 0, st, ws56, out, W.w_deconv, W.b_deconv, planes, h, w);
  else
    throw Error(SS4K_EINVAL, "FSRCNN: scale must be 2 or 4");
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
