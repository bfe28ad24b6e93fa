// FSRCNN forward (reference src/upscale/model/fsrcnn/model.py:14-62) on fp32 planes.
// Channel depths are 1 / 56 / 12, far too shallow for a dense MFMA contraction, so these are
// direct convolutions on the vector ALUs: one thread per low-resolution pixel holds every output
// channel in registers, weights are wave-uniform (scalar loads), activations are stored as groups of
// four channels, group-major ([C/4][pixel][4] fp32), so a wave's 16-byte loads/stores of one group are
// one contiguous kilobyte whatever the channel count.  Layers are fused where no halo is needed:
//   head  = conv5x5(1->56)+PReLU -> conv1x1(56->12)+PReLU        (56-wide map never leaves registers)
//   map   = conv3x3(12->12)+PReLU                                 (x4)
//   tail  = conv1x1(12->56)+PReLU -> ConvTranspose 9x9 stride s   (fused, exact-fp32 MFMA: k_fs_tail)
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

// conv3x3 12->12 + PReLU.  Each thread computes R vertically adjacent pixels of one column (lanes stay
// on consecutive x: every load is a coalesced kilobyte) and walks the R+2 input rows once: a loaded
// 12-channel position feeds up to three output rows, so a pixel costs (R+2)*3/R position loads
// instead of 9 (the one-pixel-per-thread form was bound by L1/L2 bandwidth, 432 B per pixel).
template <int R>
__global__ __launch_bounds__(256) void k_fs_map(const float* __restrict__ in, float* __restrict__ out,
                                                const float* __restrict__ wm, const float* __restrict__ bm,
                                                const float* __restrict__ am, int planes, int h, int w) {
  const int hb = (h + R - 1) / R;  // row blocks per plane
  const size_t nthreads = (size_t)planes * hb * w, total = (size_t)planes * h * w;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= nthreads) return;
  const int x = i % w, yb = (i / w) % hb, y0 = yb * R;
  const size_t pbase = (i / ((size_t)w * hb)) * (size_t)h * w;
  float s[R][12];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int j = 0; j < 12; ++j) s[r][j] = bm[j];
#pragma unroll
  for (int ir = 0; ir < R + 2; ++ir) {
    const int yy = y0 + ir - 1;
    if (yy < 0 || yy >= h) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = x + kx - 1;
      if (xx < 0 || xx >= w) continue;
      const float4* p = reinterpret_cast<const float4*>(in) + pbase + (size_t)yy * w + xx;
      float v[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) { const float4 t = p[q * total]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ky = ir - r;  // input row ir is tap row ky of output row r
        if (ky < 0 || ky > 2) continue;
#pragma unroll
        for (int c = 0; c < 12; ++c)
#pragma unroll
          for (int j = 0; j < 12; ++j) s[r][j] = fmaf(wm[((ky * 3 + kx) * 12 + c) * 12 + j], v[c], s[r][j]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (y0 + r >= h) break;
    float4* dst = reinterpret_cast<float4*>(out) + pbase + (size_t)(y0 + r) * w + x;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      dst[q * total] = make_float4(prelu(s[r][4 * q], am[4 * q]), prelu(s[r][4 * q + 1], am[4 * q + 1]),
                                   prelu(s[r][4 * q + 2], am[4 * q + 2]), prelu(s[r][4 * q + 3], am[4 * q + 3]));
  }
}

// ---------------------------------------------------------------------------------------------
// Fused tail: expand (conv1x1 12->56 + PReLU) and the 9x9 stride-S ConvTranspose in one kernel, on
// the matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32), so the 56-channel tensor (224 B per LR
// pixel, the largest intermediate of the network) never exists in memory.
//
// Scatter form of the transposed conv: every LR pixel q spreads T[q][ky][kx] = sum_c E[q][c]*Wd[ky][kx][c]
// onto output (S*qy + ky - 4, S*qx + kx - 4).  Per wave and LR row, for 32 pixels:
//   E'[64 ch][32 px] = We[64 x 12] * X[12 x 32]          12 MFMAs   (+ bias, PReLU, in registers)
//   T [96 taps][32 px] = Wd[96 x 56] * E'[56 x 32]       84 MFMAs   (the accumulators of the first
//                                                         product ARE the B operand of the second:
//                                                         same lane = same pixel, no data movement)
// then the 81 real taps are added into a 16-row ring of output rows in LDS.  A workgroup (4 waves =
// 128 LR columns of one row, 124 interior + 2 halo each side) marches down a band of rows; after LR
// row y the output rows S*y-4 .. S*y-4+S-1 are complete and leave for HBM.  Deterministic: within a
// wave-instruction no two lanes touch the same ring word (half-waves own even / odd kernel rows),
// adjacent waves add in separate barrier-separated phases, rows are visited in order.
typedef float f32x16v __attribute__((ext_vector_type(16)));
constexpr int FS_CI = 124;  // interior LR columns of a strip

template <int S>
__global__ __launch_bounds__(256) void k_fs_tail(const float* __restrict__ in12, float* __restrict__ out,
                                                 const float* __restrict__ we, const float* __restrict__ be,
                                                 const float* __restrict__ ae, const float* __restrict__ wd, float bias,
                                                 int planes, int h, int w, int bands) {
  constexpr int RW = S * 128 + 8;
  extern __shared__ float fs_lds[];
  float* wd_pk = fs_lds;          // [3 tap blocks][2 channel blocks][16 k-steps][64 lanes]  A operands of T
  float* we_pk = wd_pk + 6144;    // [2 channel blocks][6 k-steps][64 lanes]                 A operands of E'
  float* bea = we_pk + 768;       // [64] bias, [64] PReLU slope of the (padded) expand channels
  float* ring = bea + 128;        // [16][RW] output rows under construction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, hh = lane >> 5;
  const int strips = (w + FS_CI - 1) / FS_CI;
  const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) % bands, plane = blockIdx.x / (strips * bands);
  const int x0 = strip * FS_CI;
  const int rpb = (h + bands - 1) / bands, ylo = band * rpb, yhi = min(h, ylo + rpb);
  if (plane >= planes || ylo >= yhi) return;

  // operand tables.  Tap slot r of block tb sits in half hs = (r>>2)&1 of the accumulator; half 0
  // carries the 45 taps with even ky, half 1 the 36 with odd ky, each in (ky, kx) order.
  for (int e = tid; e < 6144; e += 256) {
    const int tb = e >> 11, b = (e >> 10) & 1, i = (e >> 6) & 15, l = e & 63;
    const int r = l & 31, hs = (r >> 2) & 1, is = (r & 3) + 4 * (r >> 3), ord = tb * 16 + is;
    const int c = 32 * b + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
    const bool ok = ord < (hs ? 36 : 45) && c < 56;
    const int ky = 2 * (ord / 9) + hs, kx = ord % 9;
    wd_pk[e] = ok ? wd[(ky * 9 + kx) * 56 + c] : 0.f;
  }
  for (int e = tid; e < 768; e += 256) {
    const int b = e / 384, s = (e >> 6) % 6, l = e & 63;
    const int ch = 32 * b + (l & 31), kin = 2 * s + (l >> 5);
    we_pk[e] = ch < 56 ? we[kin * 56 + ch] : 0.f;
  }
  if (tid < 64) { bea[tid] = tid < 56 ? be[tid] : 0.f; bea[64 + tid] = tid < 56 ? ae[tid] : 1.f; }
  for (int e = tid; e < 16 * RW; e += 256) ring[e] = 0.f;
  __syncthreads();

  const size_t plane_px = (size_t)h * w, total = (size_t)planes * plane_px;
  const int OW = S * w, OH = S * h;
  float* oplane = out + (size_t)plane * OH * OW;
  const int px = x0 - 2 + 32 * wave + p;
  const bool col_ok = px >= 0 && px < w;
  // the 12 input channels of this lane's pixel, fetched one row ahead
  auto load_x = [&](int yy, float4& a0, float4& a1, float4& a2) {
    a0 = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (yy >= 0 && yy < h) {
      const float4* src = reinterpret_cast<const float4*>(in12) + (size_t)plane * plane_px + (size_t)yy * w + min(max(px, 0), w - 1);
      a0 = src[0]; a1 = src[total]; a2 = src[2 * total];
    }
  };
  float4 n0, n1, n2;
  load_x(ylo - 2, n0, n1, n2);
  for (int y = ylo - 2; y < yhi + 2; ++y) {
    const bool row_ok = y >= 0 && y < h;  // wave-uniform
    f32x16v T[3];
    const float4 g0 = n0, g1 = n1, g2 = n2;
    load_x(y + 1, n0, n1, n2);
    if (row_ok) {
      // k-step s of the first product contracts input channels 2s (half 0) and 2s+1 (half 1)
      const float xb[6] = {hh ? g0.y : g0.x, hh ? g0.w : g0.z, hh ? g1.y : g1.x,
                           hh ? g1.w : g1.z, hh ? g2.y : g2.x, hh ? g2.w : g2.z};
      f32x16v E[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int i = 0; i < 16; ++i) E[b][i] = bea[32 * b + (i & 3) + 8 * (i >> 2) + 4 * hh];
#pragma unroll
        for (int s = 0; s < 6; ++s) E[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(we_pk[(b * 6 + s) * 64 + lane], xb[s], E[b], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) E[b][i] = prelu(E[b][i], bea[64 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * hh]);
      }
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[tb][i] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < (b ? 12 : 16); ++i)  // channels 32 + row < 56 <=> i < 12
            T[tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wd_pk[((tb * 2 + b) * 16 + i) * 64 + lane], E[b][i], T[tb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int phase = 0; phase < 2; ++phase) {
      if ((wave & 1) == phase && row_ok) {
        // Plain read-add-write (LDS float atomics are ~6x slower here), S consecutive kx per lane as
        // one 8/16-byte word: lanes then touch consecutive words (no bank conflicts).  Two taps can
        // meet on a ring word only across lanes with equal kernel row and kx congruent mod S, so
        // the kx groups {0..S-1}, {S..2S-1}, ... of all kernel rows are rounds of independent
        // updates: reads together, then writes; rounds stay ordered (the compiler cannot move the
        // LDS writes of one round past the reads of the next).
        typedef float fvS __attribute__((ext_vector_type(S)));
        const int cbase = S * (32 * wave + p);
        constexpr int NG = 8 / S;  // full groups; kx = 8 is the last, single-tap round
#pragma unroll
        for (int g = 0; g <= NG; ++g) {
          fvS v[5]; int off[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            const bool ok = col_ok && (hh == 0 || j < 4);
            const int rr = (S * y + 2 * j + hh - 4 + 64) & 15;
            off[j] = ok ? rr * RW + cbase + S * g : -1;
            if (g < NG) {
              v[j] = ok ? *reinterpret_cast<const fvS*>(&ring[off[j]]) : fvS(0.f);
#pragma unroll
              for (int e = 0; e < S; ++e) { const int ord = 9 * j + S * g + e; v[j][e] += T[ord >> 4][ord & 15]; }
            } else {
              const int ord = 9 * j + 8;
              v[j][0] = ok ? ring[off[j]] + T[ord >> 4][ord & 15] : 0.f;
            }
          }
#pragma unroll
          for (int j = 0; j < 5; ++j)
            if (off[j] >= 0) {
              if (g < NG) *reinterpret_cast<fvS*>(&ring[off[j]]) = v[j];
              else ring[off[j]] = v[j][0];
            }
        }
      }
      __syncthreads();
    }
    // output rows S*y-4 .. S*y-4+S-1 are final (both phases are behind a barrier): hand their
    // interior over (+ bias) and clear the whole ring rows.  No barrier after this pass: the next
    // LR row adds to rows S*y+S-4 and up, and these ring slots come back only 8/S rows later.
    for (int e = tid; e < S * RW; e += 256) {
      const int r = e / RW, c = e - r * RW;
      const int Y = S * y - 4 + r, X = S * x0 + c - 2 * S - 4;
      float* q = &ring[((Y + 64) & 15) * RW + c];
      if (c >= 2 * S + 4 && c < 2 * S + 4 + S * FS_CI && Y >= S * ylo && Y < S * yhi && X < OW)
        oplane[(size_t)Y * OW + X] = *q + bias;
      *q = 0.f;
    }
  }
}

void fsrcnn_forward(ss4k_ctx* ctx, const FsrcnnWeights& W, int factor, const float* in, float* out, int planes, int h,
                    int w, float* ws12a, float* ws12b, hipStream_t st) {
  const size_t total = (size_t)planes * h * w;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_fs_head, grid, block, 0, st, in, ws12a, W.w_feat, W.b_feat, W.a_feat, W.w_shrink, W.b_shrink,
                     W.a_shrink, planes, h, w);
  constexpr int FS_MAP_R = 4;
  const dim3 mgrid((unsigned)(((size_t)planes * ((h + FS_MAP_R - 1) / FS_MAP_R) * w + 255) / 256));
  float* cur = ws12a; float* nxt = ws12b;
  for (int l = 0; l < 4; ++l) {
    hipLaunchKernelGGL(k_fs_map<FS_MAP_R>, mgrid, block, 0, st, cur, nxt, W.w_map[l], W.b_map[l], W.a_map[l], planes, h, w);
    std::swap(cur, nxt);
  }
  const int strips = (w + FS_CI - 1) / FS_CI;
  // one round of workgroups at three per CU; every band re-does 4 halo rows
  const int bands = std::max(1, std::min((h + 15) / 16, 3 * ctx->num_cu / std::max(1, planes * strips)));
  const dim3 tgrid((unsigned)(planes * bands * strips));
  auto launch_tail = [&](auto kern, int S) {
    const size_t lds = (size_t)(6144 + 768 + 128 + 16 * (S * 128 + 8)) * 4;
    hipLaunchKernelGGL(kern, tgrid, block, lds, st, cur, out, W.w_expand, W.b_expand, W.a_expand, W.w_deconv, W.b_deconv,
                       planes, h, w, bands);
  };
  if (factor == 2) launch_tail(k_fs_tail<2>, 2);
  else if (factor == 4) launch_tail(k_fs_tail<4>, 4);
  else throw Error(SS4K_EINVAL, "FSRCNN: scale must be 2 or 4");
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
