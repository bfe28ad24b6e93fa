// FSRCNN forward (reference src/upscale/model/fsrcnn/model.py:14-62) on fp32 planes.
// Three arithmetic modes (fsrcnn_forward): fp32-grade on the fp16 matrix cores with hi/lo-split operands (default), plain fp16
// operands (an SS4K_F16 model), and the exact-fp32 kernels this file started with (SS4K_MODEL_FS_EXACT; the A/B reference).
// The exact kernels: channel depths are 1 / 56 / 12, so these are direct convolutions on the vector ALUs (head, mapping) and
// fp32 MFMA (tail): one thread per low-resolution pixel holds every output channel in registers, weights are wave-uniform
// (scalar loads).  Activations between the stages are groups of four channels, group-major ([C/4][pixel][4]; fp32, or fp16 in
// fp16 mode), so a wave's loads/stores of one group are contiguous whatever the channel count.  Layers are fused where no halo is needed:
//   head  = conv5x5(1->56)+PReLU -> conv1x1(56->12)+PReLU        (56-wide map never leaves registers)
//   map   = conv3x3(12->12)+PReLU                                 (x4)
//   tail  = conv1x1(12->56)+PReLU -> ConvTranspose 9x9 stride s   (fused, exact-fp32 MFMA: k_fs_tail)
#include "common.h"
#include "glue.h"
#include "conv_tile.h"
#include <cstdio>
#include <vector>

namespace ss4k {

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }
// PReLU of the matrix-core modes in ONE full-rate instruction.  PReLU(x) = a x + b |x| with a = (1 + s) / 2, b = (1 - s) / 2; a model of those
// modes has the weights and the bias that PRODUCE a channel scaled by its a when it is built (models.cpp; s > -1 + 1/8 for every channel, else
// the exact kernels), so the accumulator holds y = a x, and PReLU(x) = y + c |y| with c = b / a = (1 - s) / (1 + s) - which is what the model's
// "slope" arrays hold in these modes.  |y| is a source modifier: v_fma_f32 at 2.1 cycles against multiply + v_max_f32 (2.1 + 3.8: v_max is a
// half-rate instruction, tools/micro/valu_rates.hip) or, on packed fp16, v_pk_mul_f16 + v_pk_max_f16 (3.6 + 3.9).  Padded channels: c = 0.
// The exact-fp32 kernels (unscaled weights, true slopes) keep prelu().
__device__ __forceinline__ float prelu_mx(float y, float c) { return __builtin_fmaf(__builtin_fabsf(y), c, y); }

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

// ---------------------------------------------------------------------------------------------
// Production tail: the same two products on the fp16 matrix rate, and the overlap-add in REGISTERS.
//
// Arithmetic.  Every fp32 operand is split x = hi + 2^-11 * lo with hi and lo BOTH fp16 (hi = x with its low 13
// mantissa bits cleared, lo = (x - hi) * 2048: exact in fp32 and at hi's magnitude, so no fp16 subnormals), and a
// product is three v_mfma_f32_32x32x16_f16: hi*hi into one fp32 accumulator, hi*lo + lo*hi into a second one that is
// scaled by 2^-11 at the end; the dropped lo*lo term is 2^-22 of |x||w|.  22 significand bits per operand and fp32
// accumulation: the result agrees with the exact-fp32 kernel above to ~5e-7 of the output peak (tools/fs_time.py,
// test_fsrcnn_split_tail_vs_exact), far inside the path's 1e-3 / 1e-4 tolerance, at 36 + 6 MFMAs of 32 cycles per 32
// pixels instead of 84 + 12 of 64.  Operands must stay inside the fp16 range (|x| < 65504; image-range networks are
// orders of magnitude below).
//
// Overlap-add.  With the matrix time gone the ring version is bound by its LDS read-add-write rounds, two barriers and
// the hand-over pass per row (~1200 vector / LDS instructions per wave and row).  Here a wave owns 32 LR columns of
// which 32 - 2*HALO are interior, and nothing is shared between waves after the operand tables are built:
//   * along x, output column S*x' + r of kernel row ky is  sum_d T[ky][S*d + r + 4] of pixel x' - d : lane x' collects
//     from lanes x' -+ 1, 2 with DPP wave shifts folded into the adds (v_add_f32_dpp wave_shl / wave_shr: 7 adds per
//     kernel row at S = 2, 5 at S = 4);
//   * along y, a lane keeps the five output rows it is still adding to (rows S*y - 4 + 2j + hh: lane half hh owns the
//     rows of its kernel-row parity) in 5 x S registers; S/2 of them are complete after each LR row and leave for HBM
//     straight from the registers (one 8 / 16-byte store per lane and row).
// Deterministic by construction (no atomics, fixed order).
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2v __attribute__((ext_vector_type(2)));
struct HiLo { f16x8v hi, lo; };
// two fp32 values -> packed fp16 hi pair and packed fp16 lo pair
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  // hi = x rounded toward zero to fp16 (in the normal range: x with 13 mantissa bits cleared, exact); lo is computed from the
  // value that LANDED in fp16, so an fp16-subnormal hi (|x| < 2^-14) loses nothing either.  |x| >= 65504 does not fit the
  // split (hi saturates): see Model::build / SS4K_MODEL_FS_EXACT.
  const fp16x2v ph = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  // (x - hi) * 2048 as fma(hi, -2048, x * 2048): both products are exact (a power of two), the difference is exact, and the fp16 hi
  // enters the fma directly (v_fma_mix_f32) - one multiply (packed over the pair) + one mixed fma per value instead of convert,
  // subtract, multiply
  const float h0 = (float)ph[0], h1 = (float)ph[1];
  const fp16x2v pl = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf(h0, -2048.f, x0 * 2048.f), __builtin_fmaf(h1, -2048.f, x1 * 2048.f));
  hi = __builtin_bit_cast(uint32_t, ph); lo = __builtin_bit_cast(uint32_t, pl);
}
// fp16 mode (desc.dtype == SS4K_F16, the precision the reference's TensorRT engine runs this network in): operands are the
// round-to-nearest fp16 of the value, one MFMA per product, fp32 accumulation; no lo part anywhere
__device__ __forceinline__ uint32_t half2_rne(float x0, float x1) {
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  typedef _Float16 h16x2v __attribute__((ext_vector_type(2)));
  const f32x2v v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2v));
}
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// fp16 mode: PReLU in fp32 on the accumulators (prelu_mx: one fma each), then ONE rounding to fp16 - rounds 3-6 converted first and took
// max(v, v s) on the packed pair (v_cvt_pk + v_pk_mul_f16 + v_pk_max_f16 = 13 cycles a pair; now 2 x 2.1 + the conversion's 5.6)
__device__ __forceinline__ uint32_t prelu_h2(float y0, float y1, float c0, float c1) { return half2_rne(prelu_mx(y0, c0), prelu_mx(y1, c1)); }
template <bool SPLIT>
__device__ __forceinline__ void pack2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  if constexpr (SPLIT) split2(x0, x1, hi, lo); else { hi = half2_rne(x0, x1); lo = 0u; }
}
__device__ __forceinline__ float dpp_shl1(float v) {   // lane i <- lane i + 1 (0 past the wave)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shr1(float v) {   // lane i <- lane i - 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

template <int S> struct FsTailGeo { static constexpr int HALO = S == 2 ? 2 : 1, CI = 32 - 2 * HALO; };

template <int S, bool SPLIT, bool OUT_HALF = false>
__global__ __launch_bounds__(256, 3) void k_fs_tail_r(const float* __restrict__ in12, float* __restrict__ out,
                                                   const float* __restrict__ we, const float* __restrict__ be,
                                                   const float* __restrict__ ae, const float* __restrict__ wd, float bias,
                                                   int planes, int h, int w, int bands, int tall_rpb) {
  constexpr int HALO = FsTailGeo<S>::HALO, CI = FsTailGeo<S>::CI, JSTEP = S / 2;
  extern __shared__ __attribute__((aligned(16))) float fs_lds[];
  // A operands, fragment order: [tap block 3][k-step 4][lane 64] x 8 fp16, hi then lo tables; expand [2][64] x 8
  uint4* wd_hi = reinterpret_cast<uint4*>(fs_lds);   // 768 fragments of 16 B
  uint4* wd_lo = wd_hi + 768;
  uint4* we_hi = wd_lo + 768;                        // 128
  uint4* we_lo = we_hi + 128;
  float* bea = reinterpret_cast<float*>(we_lo + 128);  // [64] bias, [64] PReLU slope of the (padded) expand channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 31, hh = lane >> 5;
  const int wstrips = (w + CI - 1) / CI, strips = (wstrips + 3) / 4;
  // Bands.  Classic (tall_rpb == 0): a workgroup is four neighbouring wave strips of one of `bands` equal bands of one plane.  Tall (round 6):
  // the unit is the WAVE (waves share nothing but the weight tables) and the planes are stacked into one image of planes * h rows cut into
  // bands of tall_rpb rows, so that the wave count can be what fills the chip's slots: 720p x2, 12 planes, are 46 wave strips - classic
  // 12 workgroup strips (two idle waves each) x 5 bands x 12 planes = 720 workgroups of 768 slots and 148 row steps each, tall 66 bands of
  // 131 rows.  A band that straddles a plane boundary is marched as two segments (4 more halo rows).
  int ws, plane, ylo, yhi, tall0 = 0, tall1 = 0;
  if (tall_rpb > 0) {
    const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(wave);
    ws = wid % wstrips;
    tall0 = (wid / wstrips) * tall_rpb; tall1 = min(planes * h, tall0 + tall_rpb);
    plane = tall0 / h; ylo = tall0 - plane * h; yhi = min(h, ylo + (tall1 - tall0));
  } else {
    const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) % bands;
    plane = blockIdx.x / (strips * bands);
    const int rpb = (h + bands - 1) / bands;
    ylo = band * rpb; yhi = min(h, ylo + rpb);
    if (plane >= planes || ylo >= yhi) return;
    ws = strip * 4 + wave;
  }

  // Tap slot r of block tb sits in half hs = (r>>2)&1 of the accumulator; half 0 carries the 45 taps with even ky,
  // half 1 the 36 with odd ky, each in (ky, kx) order (as k_fs_tail).  k = 8*kq + j of k-step s is channel
  // 32*(s>>1) + (i&3) + 8*(i>>2) + 4*kq with i = 8*(s&1) + j: the channel that accumulator register i of block s>>1
  // holds in lane half kq, so the first product's accumulators are the second one's B operand as they stand.
  for (int e = tid; e < 768; e += 256) {
    const int tb = e >> 8, s4 = (e >> 6) & 3, l = e & 63;
    const int r = l & 31, kq = l >> 5, hs = (r >> 2) & 1, is = (r & 3) + 4 * (r >> 3), ord = tb * 16 + is;
    const int ky = 2 * (ord / 9) + hs, kx = ord % 9;
    uint32_t vh[4], vl[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      float v[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 8 * (s4 & 1) + j + t, c = 32 * (s4 >> 1) + (i & 3) + 8 * (i >> 2) + 4 * kq;
        v[t] = (ord < (hs ? 36 : 45) && c < 56) ? wd[(ky * 9 + kx) * 56 + c] : 0.f;
      }
      pack2<SPLIT>(v[0], v[1], vh[j >> 1], vl[j >> 1]);
    }
    wd_hi[e] = make_uint4(vh[0], vh[1], vh[2], vh[3]); wd_lo[e] = make_uint4(vl[0], vl[1], vl[2], vl[3]);
  }
  if (tid < 128) {
    const int b = tid >> 6, l = tid & 63, ch = 32 * b + (l & 31), kq = l >> 5;
    uint32_t vh[4], vl[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      float v[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kin = 8 * kq + j + t;
        v[t] = (ch < 56 && kin < 12) ? we[kin * 56 + ch] : 0.f;
        if (ch < 56 && kin == 12) v[t] = be[ch];   // input slot 12 is the constant 1 (0 outside the image): it carries the expand bias
      }
      pack2<SPLIT>(v[0], v[1], vh[j >> 1], vl[j >> 1]);
    }
    we_hi[tid] = make_uint4(vh[0], vh[1], vh[2], vh[3]); we_lo[tid] = make_uint4(vl[0], vl[1], vl[2], vl[3]);
  }
  if (tid < 64) { bea[tid] = tid < 56 ? be[tid] : 0.f; bea[64 + tid] = tid < 56 ? ae[tid] : 0.f; }   // (PReLU constants c, prelu_mx: 0 = identity)
  __syncthreads();
  // ws: this wave's strip of CI interior columns; nothing below synchronises
  if (ws >= wstrips || (tall_rpb > 0 && tall0 >= tall1)) return;

  // fp16 mode: PReLU constants of the expand channels this lane's accumulators hold (registers instead of LDS reads)
  float slp[2][SPLIT ? 1 : 16];
  if constexpr (!SPLIT) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) slp[b][i] = bea[64 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * hh];
  }
  const size_t plane_px = (size_t)h * w, total = (size_t)planes * plane_px;
  const int OW = S * w, OH = S * h;
  const int px = ws * CI - HALO + p;
  const bool col_ok = px >= 0 && px < w;
  const bool interior = p >= HALO && p < 32 - HALO && px < w;
  // this lane's 8 input channels of its pixel (half 0: channels 0-7, half 1: 8-11 + zeros), fetched PF rows ahead into a register ring.
  // (fp16 mode: the 12-channel tensor is fp16, 8 bytes per pixel and group; .x/.y of a0 and a1 carry the packed pairs.)  Every load is
  // UNCONDITIONAL - address clamped into the image, the value replaced where it is used - and several rows are in flight: with one row
  // ahead and the load under a branch (rounds 2-5) the fp16 mode, whose row is 14 MFMAs, spent 63 % of its wave cycles in s_waitcnt
  // (SQ_WAIT_ANY in the round's first counter pass, profiles/NOTES_r06.md 6) - a row is shorter than a trip to HBM.
  // fp16 mode: FIVE rows per trip of the unrolled loop - the five output rows under construction then rotate by INDEX (row j of sub-step u
  // is register set (j + JSTEP u) % 5: a compile-time constant) instead of by ten register moves per row
  constexpr int PF = SPLIT ? (S == 2 ? 3 : 2) : 5;   // (fp32-grade: three | two rows in flight fit the 170 registers of three workgroups per CU)
  constexpr bool ROT = PF == 5;
  const int pxc = min(max(px, 0), w - 1);
  const int g0i = hh ? 2 : 0, g1i = hh ? 2 : 1;   // half 1 reads group 2 twice (the same line) instead of branching
  auto load_x = [&](int yy, float4& a0, float4& a1) {
    const size_t at = (size_t)plane * plane_px + (size_t)min(max(yy, 0), h - 1) * w + pxc;
    if constexpr (SPLIT) {
      const float4* src = reinterpret_cast<const float4*>(in12) + at;
      a0 = src[g0i * total]; a1 = src[g1i * total];
    } else {
      const float2* src = reinterpret_cast<const float2*>(in12) + at;
      const float2 u = src[g0i * total], v = src[g1i * total];
      a0 = make_float4(u.x, u.y, 0.f, 0.f); a1 = make_float4(v.x, v.y, 0.f, 0.f);
    }
  };
  typedef float fvS __attribute__((ext_vector_type(S)));
  constexpr float LO = 1.f / 2048.f;
  const f32x16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
 for (;;) {   // one segment = rows [ylo, yhi) of `plane` (classic bands: exactly one)
  float* oplane = out + (size_t)plane * OH * OW;
  fvS V[5];   // output rows S*y - 4 + 2j + hh under construction, this lane's S columns of each
#pragma unroll
  for (int j = 0; j < 5; ++j) V[j] = fvS(0.f);
  float4 n0[PF], n1[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) load_x(ylo - 2 + u, n0[u], n1[u]);
  for (int yb = ylo - 2; yb < yhi + 2; yb += PF) {
#pragma unroll
   for (int u = 0; u < PF; ++u) {
    const int y = yb + u;
    if (y >= yhi + 2) break;   // wave-uniform
    const bool row_ok = y >= 0 && y < h;  // wave-uniform
    float4 g0 = n0[u], g1 = n1[u];
    load_x(y + PF, n0[u], n1[u]);
    if (!col_ok) { g0 = make_float4(0.f, 0.f, 0.f, 0.f); g1 = g0; }   // (a column outside the image is zero padding; a row outside is skipped)
    if (hh) g1 = make_float4(0.f, 0.f, 0.f, 0.f);                      // half 1: channels 8-11 only
    if (row_ok) {
      uint4 xh, xl;
      if constexpr (SPLIT) {
        split2(g0.x, g0.y, xh.x, xl.x); split2(g0.z, g0.w, xh.y, xl.y); split2(g1.x, g1.y, xh.z, xl.z); split2(g1.z, g1.w, xh.w, xl.w);
        if (hh == 1 && col_ok) xh.z = 0x00003c00u;   // slot 12 = 1 (lo part 0): carries the expand bias; a column outside the image stays all zero
      } else {
        xh = make_uint4(__float_as_uint(g0.x), __float_as_uint(g0.y), __float_as_uint(g1.x), __float_as_uint(g1.y));
        if (hh == 1 && col_ok) xh.z = 0x00003c00u;   // slot 12 = 1: carries the expand bias; a column outside the image stays all zero
        xl = make_uint4(0u, 0u, 0u, 0u);
      }
      const f16x8v bxh = __builtin_bit_cast(f16x8v, xh), bxl = __builtin_bit_cast(f16x8v, xl);
      // E' = PReLU(We * X + be), split again: B operands of the four k-steps of the second product
      uint4 ebh[4], ebl[4];
      if constexpr (!SPLIT) {
        // one MFMA per block, bias through the constant-1 slot (so E' of a column outside the image is PReLU(0) = 0), PReLU on packed fp16
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f16x8v ah = __builtin_bit_cast(f16x8v, we_hi[b * 64 + lane]);
          const f32x16v E1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bxh, zero16, 0, 0, 0);
          uint32_t e[8];
#pragma unroll
          for (int i = 0; i < 16; i += 2) e[i >> 1] = prelu_h2(E1[i], E1[i + 1], slp[b][i], slp[b][i + 1]);
          ebh[2 * b] = make_uint4(e[0], e[1], e[2], e[3]); ebh[2 * b + 1] = make_uint4(e[4], e[5], e[6], e[7]);
          ebl[2 * b] = ebl[2 * b + 1] = make_uint4(0u, 0u, 0u, 0u);
        }
      } else
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        // (the bias arrives through the constant-1 slot of the input: accumulators start from the inline constant 0, and a column
        // outside the image - all-zero input - gives PReLU(0) = 0 without a select)
        const f16x8v ah = __builtin_bit_cast(f16x8v, we_hi[b * 64 + lane]), al = __builtin_bit_cast(f16x8v, we_lo[b * 64 + lane]);
        f32x16v E1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bxh, zero16, 0, 0, 0);
        f32x16v E2 = zero16;
        if constexpr (SPLIT) {
          E2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bxl, zero16, 0, 0, 0);
          E2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bxh, E2, 0, 0, 0);
        }
        float ev[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 sl = *reinterpret_cast<const float4*>(&bea[64 + 32 * b + 8 * k + 4 * hh]);
          const float sv[4] = {sl.x, sl.y, sl.z, sl.w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float v = SPLIT ? fmaf(E2[4 * k + t], LO, E1[4 * k + t]) : E1[4 * k + t];
            ev[4 * k + t] = prelu_mx(v, sv[t]);
          }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint32_t hq[4], lq[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) pack2<SPLIT>(ev[8 * q + 2 * t], ev[8 * q + 2 * t + 1], hq[t], lq[t]);
          ebh[2 * b + q] = make_uint4(hq[0], hq[1], hq[2], hq[3]); ebl[2 * b + q] = make_uint4(lq[0], lq[1], lq[2], lq[3]);
        }
      }
      f32x16v T[3];
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        f32x16v T1 = zero16, T2 = zero16;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const f16x8v ah = __builtin_bit_cast(f16x8v, wd_hi[(tb * 4 + s4) * 64 + lane]), al = __builtin_bit_cast(f16x8v, wd_lo[(tb * 4 + s4) * 64 + lane]);
          const f16x8v bh = __builtin_bit_cast(f16x8v, ebh[s4]), bl = __builtin_bit_cast(f16x8v, ebl[s4]);
          T1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, T1, 0, 0, 0);
          if constexpr (SPLIT) {
            T2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, T2, 0, 0, 0);
            T2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, T2, 0, 0, 0);
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) T[tb][i] = SPLIT ? fmaf(T2[i], LO, T1[i]) : T1[i];
      }
      // horizontal overlap-add in registers, then into the rows under construction.  Stage by stage ACROSS the five rows, so that every wave
      // shift (folded into its add: v_add_f32_dpp) reads a register written ten or more instructions earlier - a DPP operand written by
      // the previous instruction costs a 2-cycle s_nop, and the row-by-row form was a chain of them
      auto tap = [&](int j, int kx) -> float { const int ord = 9 * j + kx; return T[ord >> 4][ord & 15]; };
      if constexpr (S == 2) {
        float a0[5], a1[5], c0[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { a0[j] = dpp_shl1(tap(j, 0)) + tap(j, 2); a1[j] = dpp_shl1(tap(j, 1)) + tap(j, 3); c0[j] = dpp_shr1(tap(j, 8)) + tap(j, 6); }
#pragma unroll
        for (int j = 0; j < 5; ++j) { a0[j] = dpp_shl1(a0[j]) + tap(j, 4); a1[j] = dpp_shl1(a1[j]) + tap(j, 5); }
#pragma unroll
        for (int j = 0; j < 5; ++j) { a0[j] = dpp_shr1(c0[j]) + a0[j]; a1[j] = dpp_shr1(tap(j, 7)) + a1[j]; }
#pragma unroll
        for (int j = 0; j < 5; ++j) { fvS& Vj = V[ROT ? (j + JSTEP * u) % 5 : j]; Vj[0] += a0[j]; Vj[1] += a1[j]; }
      } else {
        float b[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) b[j][e] = dpp_shl1(tap(j, e)) + tap(j, 4 + e);
#pragma unroll
        for (int j = 0; j < 5; ++j) b[j][0] = dpp_shr1(tap(j, 8)) + b[j][0];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          fvS& Vj = V[ROT ? (j + JSTEP * u) % 5 : j];
#pragma unroll
          for (int e = 0; e < 4; ++e) Vj[e] += b[j][e];
        }
      }
    }
    // rows j < S/2 are complete: they leave from the registers, the others move up
#pragma unroll
    for (int j = 0; j < JSTEP; ++j) {
      const int Y = S * y - 4 + 2 * j + hh;
      if (interior && Y >= S * ylo && Y < S * yhi) {
        fvS o = V[ROT ? (j + JSTEP * u) % 5 : j];
#pragma unroll
        for (int e = 0; e < S; ++e) o[e] += bias;
        if constexpr (OUT_HALF) {   // the service's fp16 HR tensor (an fp16 model): S halves per lane and row
          _Float16* oh = reinterpret_cast<_Float16*>(out) + (size_t)plane * OH * OW + (size_t)Y * OW + (size_t)S * px;
          typedef _Float16 hvS __attribute__((ext_vector_type(S)));
          hvS ohv;
#pragma unroll
          for (int e = 0; e < S; ++e) ohv[e] = (_Float16)o[e];
          *reinterpret_cast<hvS*>(oh) = ohv;
        } else
        *reinterpret_cast<fvS*>(&oplane[(size_t)Y * OW + (size_t)S * px]) = o;
      }
    }
    if constexpr (ROT) {
#pragma unroll
      for (int j = 0; j < JSTEP; ++j) V[(j + JSTEP * u) % 5] = fvS(0.f);   // the rows that left start again as the newest ones
    } else {
#pragma unroll
      for (int j = 0; j < 5; ++j) V[j] = j + JSTEP < 5 ? V[j + JSTEP] : fvS(0.f);
    }
   }
  }
  if (tall_rpb <= 0) break;
  tall0 += yhi - ylo;
  if (tall0 >= tall1) break;
  plane = tall0 / h; ylo = 0; yhi = min(h, tall1 - tall0);   // the rest of the band: the first rows of the next plane
 }
}

// ---------------------------------------------------------------------------------------------
// fp16-mode head: conv5x5(1->56)+PReLU -> conv1x1(56->12)+PReLU on v_mfma_f32_32x32x16_f16, fp16 operands, fp32 accumulation.
//
// A wave marches down a band of rows of a strip of 128 columns, on its own (a private ring of 8 input rows in LDS, fp16, zero
// outside the image: no barriers).  First product, per 32 pixels: D[32 couts x 32 px] += A[32 x 16] * B[16 x 32 px], K = kernel row
// dy (6 slots, 5 real) x 8 consecutive input columns: lane (n, kq) of K-step s supplies input row y - 2 + 2s + kq, columns
// P + 2n - 2 .. P + 2n + 5 - ONE 16-byte window (4-byte aligned) that serves pixel P + 2n (taps in slots 0-4) AND pixel P + 2n + 1
// (slots 1-5): two weight operands (shift 0 / 1), one pixel operand.  The bias rides in the spare K slot (s = 2, kq = 1 is
// kernel row 5, which does not exist): its pixel operand is the constant 1 and its weight the bias; padded cout 56 has "bias"
// 1 there, so the activation map carries a constant-1 channel that feeds the second product its bias the same way.
// PReLU runs on packed fp16 (v_pk_max / min / fma_f16: 2 instructions per value with the conversion), and the packed activations
// ARE the second product's B operand (same lane = same pixel; the shrink weights are permuted to the accumulator order, as in
// the tail): D2[32 (12 real) x 32 px] = Ws[32 x 64] * E[64 x 32].  Output: fp16, group-major [3][pixel][4].
constexpr int FH_COLS = 128, FH_RING = 8, FH_ROWH = 136 + 8;   // ring row: columns X0 - 2 .. X0 + 133 (+ pad), fp16


// SPLIT: the same structure at fp32 accuracy (the default, fp32-grade mode): every operand hi + 2^-11 lo, three MFMAs per product (as the
// mapping stage and the tail), two fp16 input rings, fp32 PReLU and re-split of the 56-channel map, fp32 output.
// U8IN (round 6): the input is the service's uint8 NHWC frame tensor itself - plane p is colour p % 3 of frame p / 3, a pixel is
// (float)byte / 255.0f, the expression of glue.hip's conversion kernel - so a job whose frames need no area resize and no denoising
// never writes or reads the fp32 colour planes (44 MB per four 720p frames, one launch)
template <bool SPLIT, bool U8IN = false>
__global__ __launch_bounds__(256, 2) void k_fs_head_m(const float* __restrict__ in, void* __restrict__ outv,
                                                      const float* __restrict__ wf, const float* __restrict__ bf,
                                                      const float* __restrict__ af, const float* __restrict__ ws,
                                                      const float* __restrict__ bs, const float* __restrict__ as,
                                                      int planes, int h, int w, int bands) {
  constexpr int NP = SPLIT ? 2 : 1;   // operand parts: hi (, lo)
  __shared__ __attribute__((aligned(16))) _Float16 ring_all[4][FH_RING][NP][FH_ROWH];
  __shared__ float slope_lds[SPLIT ? 2 * 2 * 16 + 2 * 8 : 1];   // SPLIT: fp32 slopes per lane half, re-read per use (registers are short)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, kq = lane >> 5;
  const int strips = (w + FH_COLS - 1) / FH_COLS;
  const int wid = blockIdx.x * 4 + wave;
  const int strip = wid % strips, band = (wid / strips) % bands, plane = wid / (strips * bands);
  const int rpb = (h + bands - 1) / bands, ylo = band * rpb, yhi = min(h, ylo + rpb);
  // U8IN: the 256 values a byte can become, (float)k / 255.0f exactly as glue.hip's conversion kernel divides (a true division is ~ 10
  // vector instructions; the loader needs three per row): a table read instead
  __shared__ float u8_lut[U8IN ? 256 : 1];
  if constexpr (U8IN) { u8_lut[tid & 255] = (float)(tid & 255) / 255.0f; if constexpr (!SPLIT) __syncthreads(); }
  if constexpr (SPLIT) {
    // [kq][block b][i]: slope of channel 32b + (i&3) + 8(i>>2) + 4kq; then [kq][i] for the shrink's rows (i&3) + 8(i>>2) + 4kq
    if (tid < 64) { const int q = tid >> 5, bb = (tid >> 4) & 1, i = tid & 15, c = 32 * bb + (i & 3) + 8 * (i >> 2) + 4 * q; slope_lds[tid] = c < 56 ? af[c] : 0.f; }
    else if (tid < 80) { const int q = (tid - 64) >> 3, i = tid & 7, c = (i & 3) + 8 * (i >> 2) + 4 * q; slope_lds[tid] = c < 12 ? as[c] : 0.f; }
    __syncthreads();
  }
  if (plane >= planes || ylo >= yhi) return;   // whole waves leave: nothing below synchronises across waves
  _Float16 (*ring)[NP][FH_ROWH] = ring_all[wave];
  const int X0 = strip * FH_COLS;

  // first-product weights: [part][shift g][K-step s][cout block b]; lane (m = n, kq): slot j is tap (dy = 2s + kq, dx = j - g).
  // SPLIT holds shift 0 only (registers: two waves per SIMD need <= 256) and derives shift 1 where it is used: the same eight fp16
  // slots moved up by one - except the bias slot's lanes (K-step 2, kq = 1), whose operand is the same for both shifts
  constexpr int NG = SPLIT ? 1 : 2;
  uint4 A1[NP][NG][3][2];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int co = 32 * b + n, dy = 2 * s3 + kq;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int dx = j - g;
          v[j] = 0.f;
          if (dy < 5) { if (dx >= 0 && dx < 5 && co < 56) v[j] = wf[(dy * 5 + dx) * 56 + co]; }
          else if (j == 0) v[j] = co < 56 ? bf[co] : co == 56 ? 1.f : 0.f;   // the bias slot (its pixel operand is the constant 1)
        }
        uint32_t ph[4], pl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pack2<SPLIT>(v[2 * j], v[2 * j + 1], ph[j], pl[j]);
        A1[0][g][s3][b] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        if constexpr (SPLIT) A1[1][g][s3][b] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      }
  // second-product weights: K-step s4, slot j is the channel accumulator register i = 8*(s4&1) + j of block s4>>1 holds in lane half kq
  uint4 A2[NP][4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = 8 * (s4 & 1) + j, ch = 32 * (s4 >> 1) + (i & 3) + 8 * (i >> 2) + 4 * kq;
      v[j] = n < 12 ? (ch < 56 ? ws[ch * 12 + n] : ch == 56 ? bs[n] : 0.f) : 0.f;
    }
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pack2<SPLIT>(v[2 * j], v[2 * j + 1], ph[j], pl[j]);
    A2[0][s4] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    if constexpr (SPLIT) A2[1][s4] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
  }
  // fp16 mode: PReLU constants (prelu_mx) of the channels this lane's accumulators hold (SPLIT: in LDS); 0 for a padded channel
  float sl1[2][SPLIT ? 1 : 16], sl2[SPLIT ? 1 : 8];
  if constexpr (!SPLIT) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c0 = 32 * b + (i & 3) + 8 * (i >> 2) + 4 * kq;
        sl1[b][i] = c0 < 56 ? af[c0] : 0.f;
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c0 = (i & 3) + 8 * (i >> 2) + 4 * kq;
      sl2[i] = c0 < 12 ? as[c0] : 0.f;
    }
  }

  const size_t plane_px = (size_t)h * w, total = (size_t)planes * plane_px;
  const float* src = in + (size_t)plane * plane_px;
  const uint8_t* src8 = reinterpret_cast<const uint8_t*>(in) + (size_t)(plane / 3) * plane_px * 3 + (plane % 3);
  // loader: ring column c is image column X0 - 2 + c; a lane moves columns lane, lane + 64 and (lane < 8) lane + 128
  auto fetch = [&](int y, float (&v)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int x = X0 - 2 + lane + 64 * k;
      const bool ok = y >= 0 && y < h && x >= 0 && x < w && (k < 2 || lane < 8);
      // (U8IN: the raw byte travels in the prefetch register - 256 = "outside the image" - and becomes a float in put(), one row later:
      //  looked up here, the table read would wait for the load it depends on and the row-ahead prefetch would be none)
      if constexpr (U8IN) v[k] = __int_as_float(ok ? (int)src8[((size_t)y * w + x) * 3] : 256);
      else v[k] = ok ? src[(size_t)y * w + x] : 0.f;
    }
  };
  auto put = [&](int y, const float (&vin)[3]) {
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if constexpr (U8IN) { const int b = __float_as_int(vin[k]); v[k] = b < 256 ? u8_lut[b & 255] : 0.f; }
      else v[k] = vin[k];
    }
    _Float16 (*row)[FH_ROWH] = ring[y & (FH_RING - 1)];
    if constexpr (SPLIT) {
      uint32_t h01, l01, h2, l2;
      split2(v[0], v[1], h01, l01); split2(v[2], 0.f, h2, l2);
      uint16_t* rh = reinterpret_cast<uint16_t*>(row[0]); uint16_t* rl = reinterpret_cast<uint16_t*>(row[1]);
      rh[lane] = (uint16_t)h01; rh[lane + 64] = (uint16_t)(h01 >> 16);
      rl[lane] = (uint16_t)l01; rl[lane + 64] = (uint16_t)(l01 >> 16);
      if (lane < 8) { rh[lane + 128] = (uint16_t)h2; rl[lane + 128] = (uint16_t)l2; }
    } else {
      row[0][lane] = (_Float16)v[0]; row[0][lane + 64] = (_Float16)v[1];
      if (lane < 8) row[0][lane + 128] = (_Float16)v[2];
    }
  };
  float pre[3];
  for (int y = ylo - 2; y < ylo + 3; ++y) { fetch(y, pre); put(y, pre); }
  fetch(ylo + 3, pre);
  const f32x16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto a1 = [&](int part, int g, int s3, int b) -> uint4 {   // the weight operand of shift g
    if constexpr (!SPLIT) return A1[part][g][s3][b];
    else {
      const uint4 v = A1[part][0][s3][b];
      if (g == 0) return v;
      uint4 r;
      r.x = v.x << 16; r.y = __builtin_amdgcn_alignbit(v.y, v.x, 16); r.z = __builtin_amdgcn_alignbit(v.z, v.y, 16); r.w = __builtin_amdgcn_alignbit(v.w, v.z, 16);
      if (s3 == 2 && kq == 1) r = v;
      return r;
    }
  };
  const uint4 one_op = make_uint4(0x00003c00u, 0u, 0u, 0u), zero_op = make_uint4(0u, 0u, 0u, 0u);   // {1.0, 0, ...}: the bias slot's pixel operand
  constexpr float LO = 1.f / 2048.f;
  for (int y = ylo; y < yhi; ++y) {
    // (row y + 3 enters the ring at the END of this iteration: the wait for its prefetched pixels then sits behind a row of MFMAs instead of
    // in front of this row's operand reads, which must follow the ring stores - may_alias below)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int P = X0 + 64 * pass;
      if (P >= w) break;   // wave-uniform
      // pixel operands: 8 columns from ring column 64*pass + 2n of rows y - 2 + 2s + kq
      uint4 B1[NP][3];
#pragma unroll
      for (int part = 0; part < NP; ++part)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          // (may_alias: the ring rows are WRITTEN through fp16 / 16-bit pointers in put(); read through a plain uint32_t pointer the compiler
          // may move these reads above those writes - type-based aliasing - and did, the moment round 6 unrolled the row loop by two for a
          // second input row in flight: NaNs.  That variant measured 3 % slower on the stage and was dropped; the annotation stays.)
          typedef uint32_t u32_any __attribute__((may_alias));
          const u32_any* p32 = reinterpret_cast<const u32_any*>(&ring[(y - 2 + 2 * s3 + kq) & (FH_RING - 1)][part][64 * pass + 2 * n]);
          B1[part][s3] = make_uint4(p32[0], p32[1], p32[2], p32[3]);
        }
      if (kq == 1) { B1[0][2] = one_op; if constexpr (SPLIT) B1[1][2] = zero_op; }
      uint32_t o[2][4];   // fp16 mode: [pixel parity][regs 0-1: channels of accumulator registers 0-3, 2-3: of registers 4-7]
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        // per cout block: first product, PReLU (+ re-split), and straight away the two K-steps of the second product that consume
        // this block's activations - only one block's operands are alive at a time
        f32x16v d2 = zero16, d2l = zero16;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          uint32_t E[NP][8];
          f32x16v acc = zero16, acc2 = zero16;
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) {
            const uint4 wh = a1(0, g, s3, b);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, wh), __builtin_bit_cast(f16x8v, B1[0][s3]), acc, 0, 0, 0);
            if constexpr (SPLIT) {
              const uint4 wl = a1(1, g, s3, b);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, wh), __builtin_bit_cast(f16x8v, B1[1][s3]), acc2, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, wl), __builtin_bit_cast(f16x8v, B1[0][s3]), acc2, 0, 0, 0);
            }
          }
          if constexpr (SPLIT) {
            const float4* slp = reinterpret_cast<const float4*>(slope_lds + 32 * kq + 16 * b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 sv = slp[q];
              const float sl[4] = {sv.x, sv.y, sv.z, sv.w};
              float ev[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) ev[t] = prelu_mx(fmaf(acc2[4 * q + t], LO, acc[4 * q + t]), sl[t]);
              split2(ev[0], ev[1], E[0][2 * q], E[1][2 * q]);
              split2(ev[2], ev[3], E[0][2 * q + 1], E[1][2 * q + 1]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; i += 2) E[0][i >> 1] = prelu_h2(acc[i], acc[i + 1], sl1[b][i], sl1[b][i + 1]);
          }
#pragma unroll
          for (int hs = 0; hs < 2; ++hs) {
            const int s4 = 2 * b + hs, o4 = 4 * hs;
            const uint4 e = make_uint4(E[0][o4], E[0][o4 + 1], E[0][o4 + 2], E[0][o4 + 3]);
            d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, A2[0][s4]), __builtin_bit_cast(f16x8v, e), d2, 0, 0, 0);
            if constexpr (SPLIT) {
              const uint4 el = make_uint4(E[1][o4], E[1][o4 + 1], E[1][o4 + 2], E[1][o4 + 3]);
              d2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, A2[0][s4]), __builtin_bit_cast(f16x8v, el), d2l, 0, 0, 0);
              d2l = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, A2[1][s4]), __builtin_bit_cast(f16x8v, e), d2l, 0, 0, 0);
            }
          }
        }
        if constexpr (SPLIT) {
          const float4* slp = reinterpret_cast<const float4*>(slope_lds + 64 + 8 * kq);
          const float4 s0 = slp[0], s1 = slp[1];
          const float sl[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          float of[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) of[i] = prelu_mx(fmaf(d2l[i], LO, d2[i]), sl[i]);
          // lane half 0 holds channel groups 0 (registers 0-3) and 2 (registers 4-7), lane half 1 group 1, of pixel P + 2n + g
          const int xg = P + 2 * n + g;
          if (xg < w) {
            float4* row = reinterpret_cast<float4*>(outv) + (size_t)plane * plane_px + (size_t)y * w + xg;
            row[(size_t)kq * total] = make_float4(of[0], of[1], of[2], of[3]);
            if (kq == 0) row[2 * total] = make_float4(of[4], of[5], of[6], of[7]);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; i += 2) o[g][i >> 1] = prelu_h2(d2[i], d2[i + 1], sl2[i], sl2[i + 1]);
        }
      }
      // lane half 0 holds channel groups 0 (registers 0-3) and 2 (registers 4-7), lane half 1 group 1, of pixels P + 2n, P + 2n + 1
      const int x = P + 2 * n;
      const size_t ga = (size_t)kq * total;
      if constexpr (SPLIT) {
        // (stored per pixel inside the loop above)
      } else {
        uint2* row = reinterpret_cast<uint2*>(outv) + (size_t)plane * plane_px + (size_t)y * w + x;
        if (x + 1 < w) {
          *reinterpret_cast<uint4*>(row + ga) = make_uint4(o[0][0], o[0][1], o[1][0], o[1][1]);
          if (kq == 0) *reinterpret_cast<uint4*>(row + 2 * total) = make_uint4(o[0][2], o[0][3], o[1][2], o[1][3]);
        } else if (x < w) {
          row[ga] = make_uint2(o[0][0], o[0][1]);
          if (kq == 0) row[2 * total] = make_uint2(o[0][2], o[0][3]);
        }
      }
    }
    put(y + 3, pre);
    fetch(y + 4, pre);
  }
}

// ---------------------------------------------------------------------------------------------
// Production mapping stage: the four conv3x3(12->12)+PReLU layers in ONE kernel on the fp16 matrix rate (hi/lo-split
// operands, three v_mfma_f32_16x16x32_f16 per product: see the tail above), intermediates never leaving the CU.
//
// A workgroup marches down a strip of 48 columns (40 interior + the 4 of halo each side that four layers need) of one band
// of rows; wave s IS layer s.  At step t layer s computes its output row t - 2s from three rows of layer s-1 (a ring
// of four rows per layer in LDS, written one step earlier) and stores it - already split into fp16 hi / lo, 16
// channel slots (12 real) per pixel = 64 bytes - into the ring of layer s+1; layer 3 stores fp32 to HBM.  One
// barrier per step, 51 KB of LDS, three workgroups per CU.
// GEMM shape per 16 pixels: D[16 cout x 16 px] += W[16 x 32] * X[32 x 16], K = 2 taps x 16 channel slots, five K-steps
// for the nine taps; the B operand of a lane is 16 bytes of one pixel of one tap, read from the ring where the
// producing layer left it (no conversion on the consumer side).  Zero padding of every layer = zero records
// outside the image (rows and columns), which is also what the never-written ring borders hold.
constexpr int FM_U = 3, FM_COLS = 16 * FM_U, FM_HALO = 4, FM_CI = FM_COLS - 2 * FM_HALO, FM_RW = FM_COLS + 2;
// A ring record is the pixel's 12 channels as fp16 hi parts (24 bytes) followed by their lo parts (24 bytes), packed (round 6; rounds 2-5:
// 16 slots each, 64 bytes - a quarter of every K-step was padding: 5 K-steps of 32 for 9 taps x 12 channels = 108).  The three taps of a
// kernel row are three records of a ring row, and K runs over the 27 four-channel pieces (8 bytes) of the three rows: 4 K-steps of 32.
// Lane (pixel n, quarter q) supplies pieces 8 ks + q (elements 0-3) and 8 ks + 4 + q (4-7) of K-step ks: two ds_read_b64 each for hi and
// lo.  Banks: a half-wave's pixels are 48 bytes = 12 banks apart (16 pixels: every multiple of 4 once), its two quarters read neighbouring
// pieces - the lower and the upper two banks of each group of four, except where the second piece is the first of the next record:
// 40 LDS cycles per unit of 16 pixels against 32 without a conflict and the 40 of the padded records' ten 16-byte reads
// (tools/costing/fm_packed_banks.py; a search over piece orders found nothing below 40).
constexpr int FM_RECB = 48, FM_LOB = 24, FM_UNITB = 16 * FM_RECB;
constexpr int FM_ROWB = FM_RW * FM_RECB, FM_STAGEB = 4 * FM_ROWB, FM_LDS = 4 * FM_STAGEB;
struct FsMapW { const float* w[4]; const float* b[4]; const float* a[4]; };
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 4) void k_fs_maps4(const float* __restrict__ in, float* __restrict__ out, const FsMapW W,
                                                     int planes, int h, int w, int bands, int tall_rpb) {
  extern __shared__ __attribute__((aligned(16))) char fm_ring[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int st = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's layer
  const int strips = (w + FM_CI - 1) / FM_CI;
  const int strip = blockIdx.x % strips;
  // bands: classic (tall_rpb == 0) `bands` equal bands per plane; tall: bands of tall_rpb rows of the stacked planes, a band that straddles
  // a plane boundary marched as two segments (as k_fs_maps4_h below: the grid is then what fills the chip's workgroup slots)
  int tall0 = 0, tall1 = 0, plane, ylo, yhi;
  if (tall_rpb > 0) {
    const int band = blockIdx.x / strips;
    tall0 = band * tall_rpb; tall1 = min(planes * h, tall0 + tall_rpb);
    if (tall0 >= tall1) return;
    plane = tall0 / h; ylo = tall0 - plane * h; yhi = min(h, ylo + (tall1 - tall0));
  } else {
    const int band = (blockIdx.x / strips) % bands;
    plane = blockIdx.x / (strips * bands);
    const int rpb = (h + bands - 1) / bands;
    ylo = band * rpb; yhi = min(h, ylo + rpb);
    if (plane >= planes || ylo >= yhi) return;
  }
  const int x0 = strip * FM_CI;

  // A operands of this wave's layer: row m = lane & 15 is the output channel; element j of lane quarter q in K-step ks is element j & 3
  // of piece 8 ks + q + 4 (j >> 2), and element e of piece p is K index 4 p + e = (tap, channel) = ((4 p + e) / 12, (4 p + e) % 12)
  uint4 ah[4], al[4];
  {
    const float* wm = W.w[st];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint32_t vh[4], vl[4];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        float v[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          const int kk = 4 * (8 * ks + q + 4 * (j >> 2)) + ((j + t2) & 3), tap = kk / 12, ch = kk - 12 * tap;
          v[t2] = (kk < 108 && n < 12) ? wm[(tap * 12 + ch) * 12 + n] : 0.f;
        }
        split2(v[0], v[1], vh[j >> 1], vl[j >> 1]);
      }
      ah[ks] = make_uint4(vh[0], vh[1], vh[2], vh[3]); al[ks] = make_uint4(vl[0], vl[1], vl[2], vl[3]);
    }
  }
  // epilogue constants of the channels this lane holds (4q .. 4q+3), and its B-operand addressing: piece p of the window of pixel n
  // (unit 0) is 8 bytes at kernel row p / 9, piece (p % 9) % 3 of record n + (p % 9) / 3 (record n: image column x0 - 4 + n - 1); pieces 27 .. 31 do not exist
  // (their weights are zero): the last real one is read instead
  f32x4v bia, slo;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = 4 * q + i; bia[i] = c < 12 ? W.b[st][c] : 0.f; slo[i] = c < 12 ? W.a[st][c] : 0.f; }
  // (the kernel row p / 9 of slot (ks, e) is the same for all four quarters except in two slots: (1, 0) - pieces 8 | 9, 10, 11 - and
  // (2, 0) - pieces 16, 17 | 18, 19: the ring row's offset is a scalar everywhere else)
  int pc_off[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int pp = min(8 * ks + q + 4 * e, 26);
      pc_off[ks][e] = (n + (pp % 9) / 3) * FM_RECB + 8 * ((pp % 9) % 3);
    }
  // where this lane's output lands in the next ring: record n + 1 of unit 0, channels 4q .. 4q+3: bytes 8 q of the hi and of the lo part
  const int wr_hi = (n + 1) * FM_RECB + 8 * q;
  const int wr_lo = wr_hi + FM_LOB;

  const size_t plane_px = (size_t)h * w, total = (size_t)planes * plane_px;
  constexpr float LO = 1.f / 2048.f;
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
  int lo_delta = FM_LOB;
  asm volatile("" : "+s"(lo_delta));
 for (int seg = 0;; ++seg) {   // one segment = rows [ylo, yhi) of `plane` (classic bands: exactly one)
  if (seg) __syncthreads();   // (every wave is done with the rings)
  for (int e = tid; e < FM_LDS / 16; e += 256) reinterpret_cast<uint4*>(fm_ring)[e] = make_uint4(0u, 0u, 0u, 0u);
  const float4* in4 = reinterpret_cast<const float4*>(in) + (size_t)plane * plane_px;
  float4* out4 = reinterpret_cast<float4*>(out) + (size_t)plane * plane_px;
  // input loader: thread tid < 144 moves channel group g = tid / 48 of ring column c = tid % 48 (+1) of one input row
  const int lg = tid / FM_COLS, lc = tid % FM_COLS, lx = x0 - FM_HALO + lc;
  const bool loader = tid < 3 * FM_COLS, lcol_ok = lx >= 0 && lx < w;
  const int ld_hi = (lc + 1) * FM_RECB + 8 * min(lg, 2);
  const int ld_lo = ld_hi + FM_LOB;
  // unconditional loads (address clamped into the image, zero selected when the row is stored): a load under a branch makes hipcc wait
  // with vmcnt(0), and the barrier below is the LDS-only one - so the fetched row really stays in flight across a step
  const size_t lsrc = (size_t)min(lg, 2) * total + (size_t)min(max(lx, 0), w - 1);
  auto load_row = [&](int r) -> float4 {   // relative row r = image row ylo - 4 + r
    return in4[lsrc + (size_t)min(max(ylo - 4 + r, 0), h - 1) * w];
  };
  auto store_row = [&](int r, float4 v) {
    if (!loader) return;
    const int y = ylo - 4 + r;
    const bool in_img = lcol_ok && y >= 0 && y < h;
    v.x = in_img ? v.x : 0.f; v.y = in_img ? v.y : 0.f; v.z = in_img ? v.z : 0.f; v.w = in_img ? v.w : 0.f;
    uint32_t h0, h1, l0, l1;
    split2(v.x, v.y, h0, l0); split2(v.z, v.w, h1, l1);
    char* row = fm_ring + (r & 3) * FM_ROWB;   // ring of layer 0
    *reinterpret_cast<uint2*>(row + ld_hi) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(row + ld_lo) = make_uint2(l0, l1);
  };
  __syncthreads();   // rings are zero
  store_row(0, load_row(0)); store_row(1, load_row(1));
  float4 nxt = load_row(2);
  __syncthreads();
  const int nsteps = (yhi - ylo) + 4 + 6;   // layer 3 reaches relative row (yhi - ylo) + 3 at step that + 6
  for (int t = 0; t < nsteps; ++t) {
    // input row t + 2 goes into layer 0's ring while rows t - 1 .. t + 1 are being read; row t + 3 is fetched
    store_row(t + 2, nxt);
    nxt = load_row(t + 3);
    const int r = t - 2 * st;   // this layer's output row (wave-uniform)
    if (r >= 0 && r <= (yhi - ylo) + 3 + (3 - st)) {
      const int y = ylo - 4 + r;
      const bool row_in = y >= 0 && y < h;
      const char* src = fm_ring + st * FM_STAGEB;
      const char* pp[4][2];   // this step's B-operand addresses of unit 0 (hi parts; lo parts FM_LOB, units FM_UNITB further)
      const int rb0 = ((r - 1) & 3) * FM_ROWB, rb1 = (r & 3) * FM_ROWB, rb2 = ((r + 1) & 3) * FM_ROWB;   // wave-uniform
      pp[0][0] = src + rb0 + pc_off[0][0]; pp[0][1] = src + rb0 + pc_off[0][1];
      pp[1][0] = src + (q == 0 ? rb0 : rb1) + pc_off[1][0]; pp[1][1] = src + rb1 + pc_off[1][1];
      pp[2][0] = src + (q < 2 ? rb1 : rb2) + pc_off[2][0]; pp[2][1] = src + rb2 + pc_off[2][1];
      pp[3][0] = src + rb2 + pc_off[3][0]; pp[3][1] = src + rb2 + pc_off[3][1];
      // the lo parts through addresses of their own, FM_LOB further by a value hipcc cannot see: with a constant it merges each hi / lo
      // pair into one ds_read2_b64, which the LDS serves at half the rate of two ds_read_b64 (8 cycles against 2 + 2: measured, the stage
      // 32 % slower than with the padded records)
      const char* pq[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 2; ++e) pq[ks][e] = pp[ks][e] + lo_delta;
      char* dst = fm_ring + (st + 1) * FM_STAGEB + (r & 3) * FM_ROWB;   // (layer 3 writes to HBM instead)
#pragma unroll
      for (int u = 0; u < FM_U; ++u) {
        f32x4v v = zero4;
        const int x = x0 - FM_HALO + 16 * u + n;
        if (row_in) {   // wave-uniform
          f32x4v d1 = bia, d2 = zero4;
          // all sixteen operand reads of the unit first (left to itself hipcc waits for every pair right before its MFMAs)
          uint4 fh[4], fl[4];
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint2 h0 = *reinterpret_cast<const uint2*>(pp[ks][0] + u * FM_UNITB), h1 = *reinterpret_cast<const uint2*>(pp[ks][1] + u * FM_UNITB);
            const uint2 l0 = *reinterpret_cast<const uint2*>(pq[ks][0] + u * FM_UNITB), l1 = *reinterpret_cast<const uint2*>(pq[ks][1] + u * FM_UNITB);
            fh[ks] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            fl[ks] = make_uint4(l0.x, l0.y, l1.x, l1.y);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8v bh = __builtin_bit_cast(f16x8v, fh[ks]), bl = __builtin_bit_cast(f16x8v, fl[ks]);
            const f16x8v wh = __builtin_bit_cast(f16x8v, ah[ks]), wl = __builtin_bit_cast(f16x8v, al[ks]);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, d2, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, d2, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = prelu_mx(fmaf(d2[i], LO, d1[i]), slo[i]);
          const int xu = x0 - FM_HALO + 16 * u;   // wave-uniform: only units that straddle the image edge mask
          if (xu < 0 || xu + 15 >= w) {
            const bool ok = x >= 0 && x < w;       // a column outside the image is zero padding for the next layer
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ok ? v[i] : 0.f;
          }
        }
        if (st < 3) {
          if (q < 3) {
            uint32_t h0, h1, l0, l1;
            split2(v[0], v[1], h0, l0); split2(v[2], v[3], h1, l1);
            *reinterpret_cast<uint2*>(dst + u * FM_UNITB + wr_hi) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + u * FM_UNITB + wr_lo) = make_uint2(l0, l1);
          }
        } else if (row_in && q < 3 && y >= ylo && y < yhi && x >= x0 && x < x0 + FM_CI && x < w) {
          out4[q * total + (size_t)y * w + x] = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    lds_barrier();   // LDS-only (conv_tile.h): the fetched row and layer 3's stores stay in flight across it
  }
  if (tall_rpb <= 0) break;
  tall0 += yhi - ylo;
  if (tall0 >= tall1) break;
  plane = tall0 / h; ylo = 0; yhi = min(h, tall1 - tall0);   // the rest of the band: the first rows of the next plane
 }
}

// ---------------------------------------------------------------------------------------------
// fp16-mode mapping stage: the same four-layer row march (wave s is layer s, four-row rings in LDS, one barrier per row),
// re-shaped so that a row costs a wave 9 MFMAs and ~60 other instructions instead of 15 + 270:
//   * records are the pixel's 12 fp16 channels = 24 bytes, packed (round 6; rounds 3-5: 16 slots = 32 bytes with a constant-1 slot that
//     carried the bias and 3 spare - a quarter of every K-step was padding);
//   * v_mfma_f32_32x32x16_f16 with the 32 rows = (pixel parity g, 16 cout slots) and the 32 columns = pixel PAIRS: K = a window of
//     four columns (x - 1 .. x + 2 of the even pixel x) x 12 channels = 48 = THREE K-steps per kernel row, 9 per row of 64 pixels
//     (12 with the padded records); K-step s of lane half hh is the 16 bytes at 32 s + 16 hh of the window's 96 contiguous bytes - a
//     window starts at record 2n = byte 48 n of the ring row, so every read is a 16-byte-aligned ds_read_b128, and at a lane stride of
//     48 bytes the 16 lanes of a b128 group cover all 64 banks (no swizzle).  The weight operand holds tap dx = c - g at window
//     column c (a two-pixel Toeplitz block), the bias is the first MFMA's C operand, and lane (n, hh) ends up with the 12 channels of
//     pixel 2n + hh: its record for the next layer, written with three 8-byte stores;
//   * layer s keeps input row R in ring slot (R + 2s) & 3, so at step t EVERY layer reads slots (t-1, t, t+1) & 3 and writes
//     slot (t+2) & 3 of the next ring: with the step loop unrolled by four all LDS offsets are immediates;
//   * PReLU on packed fp16.
// NU units of 64 columns per workgroup strip (a wave computes NU x 32 pixel pairs per row: NU independent accumulators, reads and
// epilogues between two barriers; 8 halo columns per 64 NU instead of per 64)
constexpr int MH_HALO = 4, MH_RECB = 24, MH_UNITB = 64 * MH_RECB;
template <int NU> struct MhGeo {
  static constexpr int COLS = 64 * NU, CI = COLS - 2 * MH_HALO, REC = COLS + 2, ROWB = REC * MH_RECB, STAGEB = 4 * ROWB, LDS = 4 * STAGEB;
};

template <bool STAMP, int NU>
__global__ __launch_bounds__(256, NU == 2 ? 3 : 4) void k_fs_maps4_h(const uint2* __restrict__ in, uint2* __restrict__ out, const FsMapW W,
                                                            int planes, int h, int w, int bands, unsigned long long* dbg, int tall_rpb = 0) {
  constexpr int MH_COLS = MhGeo<NU>::COLS, MH_CI = MhGeo<NU>::CI, MH_ROWB = MhGeo<NU>::ROWB, MH_STAGEB = MhGeo<NU>::STAGEB, MH_LDS = MhGeo<NU>::LDS;
  extern __shared__ __attribute__((aligned(16))) char mh_ring[];
  // dev build, STAMP: cycles of this wave per phase (s_memtime): [0] loader, [1] operand reads + MFMAs, [2] epilogue + stores, [3] barrier
  unsigned long long ph[4] = {0, 0, 0, 0}, tlast = 0, rt0 = 0, ct0 = 0;
  if constexpr (STAMP) { rt0 = __builtin_amdgcn_s_memrealtime(); ct0 = __builtin_amdgcn_s_memtime(); }
  auto stamp = [&](int k) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (k >= 0) ph[k] += t - tlast;
      tlast = t;
    }
  };
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
  const int st = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's layer
  const int strips = (w + MH_CI - 1) / MH_CI;
  const int strip = blockIdx.x % strips;
  // Bands.  Classic (tall_rpb == 0): `bands` equal bands per plane.  Tall (round 6): the planes stacked into one image of planes * h rows
  // cut into bands of tall_rpb rows - so that the workgroup count can be what fills the chip's slots (12 planes x 23 strips x 3 bands are
  // 828 workgroups for 1024 slots; 44 tall bands x 23 are 1012) - and a band that straddles a plane boundary is marched as two segments
  // (the four-layer pipeline drains and refills at the boundary: 14 more steps for that workgroup).
  int tall0 = 0, tall1 = 0, plane, ylo, yhi;
  if (tall_rpb > 0) {
    const int band = blockIdx.x / strips;
    tall0 = band * tall_rpb; tall1 = min(planes * h, tall0 + tall_rpb);
    if (tall0 >= tall1) return;
    plane = tall0 / h; ylo = tall0 - plane * h; yhi = min(h, ylo + (tall1 - tall0));
  } else {
    const int band = (blockIdx.x / strips) % bands;
    plane = blockIdx.x / (strips * bands);
    const int rpb = (h + bands - 1) / bands;
    ylo = band * rpb; yhi = min(h, ylo + rpb);
    if (plane >= planes || ylo >= yhi) return;
  }
  const int x0 = strip * MH_CI;

  // weights of this wave's layer: row m = (co & 3) + 8 * (co >> 2) + 4 * g (the accumulator register order, so lane half g holds
  // pixel g's couts 0..15 in registers 0..15); K-step (dy, s): element 16 s + 8 hh + j of the window = column c, channel ch
  uint4 A[3][3];
  f32x16v bias16;   // the first MFMA's C operand: register i of either lane half is cout i
  {
    const float* wm = W.w[st]; const float* bm = W.b[st];
    const int g = (n >> 2) & 1, co = (n & 3) + 4 * (n >> 3);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * s + 8 * hh + j, c = k / 12, ch = k - 12 * c, dx = c - g;
          v[j] = (co < 12 && dx >= 0 && dx < 3) ? wm[((dy * 3 + dx) * 12 + ch) * 12 + co] : 0.f;
        }
        A[dy][s] = make_uint4(half2_rne(v[0], v[1]), half2_rne(v[2], v[3]), half2_rne(v[4], v[5]), half2_rne(v[6], v[7]));
      }
#pragma unroll
    for (int i = 0; i < 16; ++i) bias16[i] = i < 12 ? bm[i] : 0.f;
  }
  float slp[12];   // PReLU constants (prelu_mx) of this wave's layer: wave-uniform
#pragma unroll
  for (int j = 0; j < 12; ++j) slp[j] = W.a[st][j];
  // pixel operand of K-step (dy, s): bytes 32 s + 16 hh of the window that starts at record 2n (record = ring column + 1: records 0 and
  // 65 are the zero borders)
  const int rd = st * MH_STAGEB + 2 * n * MH_RECB + 16 * hh;
  // this lane's output pixel of unit un: ring column 2n + hh + 64 un, its record in the next layer's ring
  const int oc = 2 * n + hh, X = x0 - MH_HALO + oc;
  const int wr = (st + 1) * MH_STAGEB + (oc + 1) * MH_RECB;
  const bool edge = x0 - MH_HALO < 0 || x0 - MH_HALO + MH_COLS > w;   // wave-uniform: this strip has columns outside the image

  const size_t plane_px = (size_t)h * w, total = (size_t)planes * plane_px;
  const uint2* src = in + (size_t)plane * plane_px;
  uint2* dst = out + (size_t)plane * plane_px;
  unsigned long long pre = 0;
  int steps_done = 0;
 for (int seg = 0;; ++seg) {   // one segment = rows [ylo, yhi) of `plane` (classic bands: exactly one)
  if (seg) { src = in + (size_t)plane * plane_px; dst = out + (size_t)plane * plane_px; __syncthreads(); }   // (every wave is done with the rings)
  for (int e = tid; e < MH_LDS / 16; e += 256) reinterpret_cast<uint4*>(mh_ring)[e] = make_uint4(0u, 0u, 0u, 0u);
  // input loader: thread tid < 192 moves channel group lg = tid / 64 (8 bytes) of ring column lc = tid % 64 (layer 3's wave has none)
  const int lg = tid >> 6, lc = tid & 63, lx = x0 - MH_HALO + lc;
  const int ld_off = (lc + 1) * MH_RECB + 8 * lg;
  // Every load is UNCONDITIONAL (address clamped into the image, the value replaced when it is stored): a load under a branch makes
  // hipcc wait with vmcnt(0), which also waits for the rows requested after it and turns the four-row prefetch into none
  const size_t lplane = (size_t)min(lg, 2) * total;
  auto load_row = [&](int r, int un) -> uint2 {   // relative row r = image row ylo - 4 + r, column lc + 64 un of the strip
    return src[lplane + (size_t)min(max(lx + 64 * un, 0), w - 1) + (size_t)min(max(ylo - 4 + r, 0), h - 1) * w];
  };
  auto store_row = [&](int r, int un, uint2 v) {   // into slot r & 3
    const int y = ylo - 4 + r, xx = lx + 64 * un;
    const bool in_img = xx >= 0 && xx < w && y >= 0 && y < h;
    uint2 o;   // (component-wise selects: selecting between whole values makes hipcc select between their addresses in scratch)
    o.x = in_img ? v.x : 0u;
    o.y = in_img ? v.y : 0u;
    if (lg < 3) *reinterpret_cast<uint2*>(mh_ring + (r & 3) * MH_ROWB + ld_off + un * MH_UNITB) = o;   // (wave-uniform)
  };
  __syncthreads();   // rings are zero
#pragma unroll
  for (int un = 0; un < NU; ++un) { store_row(0, un, load_row(0, un)); store_row(1, un, load_row(1, un)); }
  // four input rows in flight: a step is shorter than a trip to HBM, so row t + 2 was requested four steps before it is stored
  uint2 nxt[NU][4];
#pragma unroll
  for (int un = 0; un < NU; ++un)
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[un][(i + 2) & 3] = load_row(i + 2, un);
  __syncthreads();
  const f32x16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nsteps = (yhi - ylo) + 4 + 6;   // layer 3 reaches relative row (yhi - ylo) + 3 at step that + 6
  const int rlast = (yhi - ylo) + 3 + (3 - st);
  if constexpr (STAMP) { if (!seg) pre = __builtin_amdgcn_s_memtime() - ct0; }
  steps_done += nsteps;
  for (int t0 = 0; t0 < nsteps; t0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      if (t >= nsteps) break;   // uniform over the workgroup
      // input row t + 2 goes into layer 0's ring (slot (t + 2) & 3) while rows t - 1 .. t + 1 are being read; row t + 6 is requested
      stamp(-1);
#pragma unroll
      for (int un = 0; un < NU; ++un) {
        store_row(t + 2, un, nxt[un][(u + 2) & 3]);
        nxt[un][(u + 2) & 3] = load_row(t + 6, un);
      }
      stamp(0);
      const int r = t - 2 * st;   // this layer's output row (wave-uniform)
      if (r >= 0 && r <= rlast) {
        const int y = ylo - 4 + r;
        const bool row_in = y >= 0 && y < h;
        uint32_t E[NU][6];
#pragma unroll
        for (int un = 0; un < NU; ++un)
#pragma unroll
          for (int j = 0; j < 6; ++j) E[un][j] = 0u;
        if (row_in) {   // wave-uniform
          f32x16v acc[NU];
          // the three pixel operands of a kernel row (of every unit) are read together, then its MFMAs issue; the next row's reads overlap
          // them (left to itself hipcc re-uses one register quad: read, wait, MFMA, nine times over)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            uint4 b[NU][3];
#pragma unroll
            for (int un = 0; un < NU; ++un)
#pragma unroll
              for (int s = 0; s < 3; ++s) b[un][s] = *reinterpret_cast<const uint4*>(mh_ring + rd + 32 * s + un * MH_UNITB + ((u + 3 + dy) & 3) * MH_ROWB);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
              for (int un = 0; un < NU; ++un)
                acc[un] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, A[dy][s]), __builtin_bit_cast(f16x8v, b[un][s]),
                                                                 (dy == 0 && s == 0) ? bias16 : acc[un], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          stamp(1);
#pragma unroll
          for (int un = 0; un < NU; ++un) {
#pragma unroll
            for (int j = 0; j < 6; ++j) E[un][j] = prelu_h2(acc[un][2 * j], acc[un][2 * j + 1], slp[2 * j], slp[2 * j + 1]);
            const int Xu = X + 64 * un;
            if (edge && !(Xu >= 0 && Xu < w)) {   // a column outside the image is zero padding for the next layer
#pragma unroll
              for (int j = 0; j < 6; ++j) E[un][j] = 0u;
            }
          }
        }
#pragma unroll
        for (int un = 0; un < NU; ++un) {
          const int ocu = oc + 64 * un, Xu = X + 64 * un;
          if (st < 3) {
#pragma unroll
            for (int q = 0; q < 3; ++q)
              *reinterpret_cast<uint2*>(mh_ring + wr + 8 * q + un * MH_UNITB + ((u + 2) & 3) * MH_ROWB) = make_uint2(E[un][2 * q], E[un][2 * q + 1]);
          } else if (row_in && y >= ylo && y < yhi && ocu >= MH_HALO && ocu < MH_HALO + MH_CI && Xu < w) {
            uint2* o = dst + (size_t)y * w + Xu;
            o[0] = make_uint2(E[un][0], E[un][1]); o[total] = make_uint2(E[un][2], E[un][3]); o[2 * total] = make_uint2(E[un][4], E[un][5]);
          }
        }
      }
      stamp(2);
      lds_barrier();   // not __syncthreads(): the prefetched rows stay in flight across it (conv_tile.h)
      stamp(3);
    }
  }
  if (tall_rpb <= 0) break;
  tall0 += yhi - ylo;
  if (tall0 >= tall1) break;
  plane = tall0 / h; ylo = 0; yhi = min(h, tall1 - tall0);   // the rest of the band: the first rows of the next plane
 }
  if constexpr (STAMP) {
    if (lane == 0 && dbg && blockIdx.x < 1024) {
      unsigned long long* o = dbg + ((size_t)blockIdx.x * 4 + st) * 8;
      for (int k = 0; k < 4; ++k) o[k] = ph[k];
      o[5] = (unsigned long long)steps_done; o[4] = pre; o[6] = __builtin_amdgcn_s_memtime() - ct0;
      o[7] = ((__builtin_amdgcn_s_memtime() - ct0) << 20) / (__builtin_amdgcn_s_memrealtime() - rt0 + 1);
    }
  }
}

void fsrcnn_forward(ss4k_ctx* ctx, const FsrcnnWeights& W, int factor, const float* in, float* out, int planes, int h,
                    int w, float* ws12a, float* ws12b, int mode, hipStream_t st, bool out_half, bool in_u8) {
  const bool exact = mode == FS_MODE_EXACT, half = mode == FS_MODE_HALF;
  SS4K_REQUIRE(!out_half || half, "FSRCNN: an fp16 output tensor is offered in fp16 mode only");
  SS4K_REQUIRE(exact || W.prelu_abs, "FSRCNN matrix-core modes: the weight blob is not scaled for the one-fma PReLU (models.cpp)");
  SS4K_REQUIRE(!in_u8 || (!exact && planes % 3 == 0), "FSRCNN: uint8 NHWC input is read by the matrix-core head only, three colour planes per frame");
  const size_t total = (size_t)planes * h * w;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  // exact (SS4K_FS_EXACT=1 when the model was built): the exact-fp32 kernels - vector-ALU mapping layers, fp32-MFMA tail with
  // the LDS ring - A/B switch and the reference of the split-precision test
  // (an fp16-split MFMA head - 16x16x32, the first product's accumulators feeding the 1x1 shrink - was built and measured:
  // 0.56 ms against this kernel's 0.62 per 12 planes of 720p; PReLU + re-splitting the 56-channel map costs ~7 vector
  // instructions per value, as many as the whole exact vector-ALU chain: not kept)
  // per-stage timing for bench.py's stage rooflines (ss4k_prof_read_kind); algorithmic FLOPs per LR pixel and plane:
  // head 2 * (25 * 56 + 56 * 12), mapping 2 * 4 * 9 * 12 * 12, tail 2 * (12 * 56 + 81 * 56)  (12 464 MAC in all)
  ProfScope prof_head(ctx, st, PROF_FS_HEAD);
  // fp16 mode and the default fp32-grade mode run the head on the matrix cores (k_fs_head_m: one | three MFMAs per product); the exact
  // mode keeps the vector-ALU kernel
  if (!exact) {
    const int hstrips = (w + FH_COLS - 1) / FH_COLS;
    // two waves (fp32-grade: one) per SIMD over the chip, bands of at least 8 rows (every band re-reads 4 halo rows)
    const int per_simd = 2;
    const int hb0 = std::max(1, std::min((h + 7) / 8, 4 * per_simd * ctx->num_cu / std::max(1, planes * hstrips)));
    const int hbands = (h + (h + hb0 - 1) / hb0 - 1) / ((h + hb0 - 1) / hb0);
    const unsigned hwaves = (unsigned)(planes * hbands * hstrips);
    if (half) {
      SS4K_REQUIRE(W.prelu_abs, "FSRCNN fp16 mode: the weight blob is not scaled for the one-fma PReLU (models.cpp)");
      auto hk = in_u8 ? &k_fs_head_m<false, true> : &k_fs_head_m<false, false>;
      hipLaunchKernelGGL(hk, dim3((hwaves + 3) / 4), block, 0, st, in, static_cast<void*>(ws12a), W.w_feat, W.b_feat,
                         W.a_feat, W.w_shrink, W.b_shrink, W.a_shrink, planes, h, w, hbands);
    }
    else {
      auto hk = in_u8 ? &k_fs_head_m<true, true> : &k_fs_head_m<true, false>;
      hipLaunchKernelGGL(hk, dim3((hwaves + 3) / 4), block, 0, st, in, static_cast<void*>(ws12a), W.w_feat, W.b_feat,
                         W.a_feat, W.w_shrink, W.b_shrink, W.a_shrink, planes, h, w, hbands);
    }
  } else
  hipLaunchKernelGGL(k_fs_head, grid, block, 0, st, in, ws12a, W.w_feat, W.b_feat, W.a_feat, W.w_shrink, W.b_shrink,
                     W.a_shrink, planes, h, w);
  prof_head.done(4144.0 * (double)total);
  ProfScope prof_map(ctx, st, PROF_FS_MAP);
  constexpr int FS_MAP_R = 4;
  const dim3 mgrid((unsigned)(((size_t)planes * ((h + FS_MAP_R - 1) / FS_MAP_R) * w + 255) / 256));
  float* cur = ws12a; float* nxt = ws12b;
  if (exact) {
    for (int l = 0; l < 4; ++l) {
      hipLaunchKernelGGL(k_fs_map<FS_MAP_R>, mgrid, block, 0, st, cur, nxt, W.w_map[l], W.b_map[l], W.a_map[l], planes, h, w);
      std::swap(cur, nxt);
    }
  } else {
    FsMapW mw;
    for (int l = 0; l < 4; ++l) { mw.w[l] = W.w_map[l]; mw.b[l] = W.b_map[l]; mw.a[l] = W.a_map[l]; }
    const int mstrips = (w + FM_CI - 1) / FM_CI;
    // at most one round of workgroups at four per CU (a second, partly filled round costs a whole march); every band
    // re-does 8 halo rows plus 6 steps of pipeline fill
    const int mbands = std::max(1, std::min((h + 31) / 32, 4 * ctx->num_cu / std::max(1, planes * mstrips)));
    if (half) {
      // strips of 64 NU columns (56 / 120 interior).  NU = 2 (round 5, VERDICT r4 item 8: two independent units of work per wave between
      // barriers, half the barriers and halo columns per pixel - at two workgroups per CU instead of four, 66.5 KB of rings each) is
      // bit-identical and 8.7 % SLOWER on the stage (0.375 against 0.345 ms per 12 planes of 720p, profiles/earlier/r05/r05_fs_nu_ab.txt): four
      // resident workgroups per CU interleave better than one wave does with itself.  Kept as a dev-library switch (SS4K_MH_NU=2).
      // Round 6, packed records (rings of 50 KB, three workgroups per CU, tall bands): 0.235 against 0.238 ms - within a percent; still the switch.
      int mh_nu = 1;
#ifdef SS4K_DEV
      if (const char* e = std::getenv("SS4K_MH_NU")) mh_nu = std::atoi(e);
#endif
      const int NUr = (mh_nu != 2 || w <= MhGeo<1>::CI) ? 1 : 2;
      const int MH_CI = NUr == 2 ? MhGeo<2>::CI : MhGeo<1>::CI, MH_LDS = NUr == 2 ? MhGeo<2>::LDS : MhGeo<1>::LDS;
      const int hs = (w + MH_CI - 1) / MH_CI;
      const int per_cu = NUr == 2 ? 3 : 4;   // resident workgroups per CU (rings of 25 | 50 KB)
      int hb = std::max(1, std::min((h + 31) / 32, per_cu * ctx->num_cu / std::max(1, planes * hs)));   // one round
#ifdef SS4K_DEV
      if (const char* e = std::getenv("SS4K_MH_BANDS")) hb = std::max(1, std::atoi(e));
#endif
#ifdef SS4K_DEV
      static const bool stamp_mode = std::getenv("SS4K_FS_STAMP") && std::getenv("SS4K_FS_STAMP")[0] == '1';
      if (stamp_mode) {   // phase cycle counters of every wave of the first 1024 workgroups, printed per layer
        static unsigned long long* dbuf = nullptr;
        if (!dbuf) SS4K_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 4 * 8 * 8));
        SS4K_HIP(hipMemsetAsync(dbuf, 0, 1024 * 4 * 8 * 8, st));
        auto kst = NUr == 2 ? &k_fs_maps4_h<true, 2> : &k_fs_maps4_h<true, 1>;
        const void* fs = reinterpret_cast<const void*>(kst);
        if (ctx->lds_attr_set.insert(fs).second) SS4K_HIP(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, MH_LDS));
        hipLaunchKernelGGL(kst, dim3((unsigned)(planes * hb * hs)), block, MH_LDS, st, reinterpret_cast<const uint2*>(cur),
                           reinterpret_cast<uint2*>(nxt), mw, planes, h, w, hb, dbuf, 0);
        SS4K_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> hbuf(1024 * 4 * 8);
        SS4K_HIP(hipMemcpy(hbuf.data(), dbuf, hbuf.size() * 8, hipMemcpyDeviceToHost));
        for (int l = 0; l < 4; ++l) {
          double acc[4] = {0, 0, 0, 0}, steps = 0, clk = 0, pre = 0, tot = 0; int nw = 0;
          for (int wg = 0; wg < 1024; ++wg) {
            const unsigned long long* o = &hbuf[((size_t)wg * 4 + l) * 8];
            if (!o[5]) continue;
            for (int k = 0; k < 4; ++k) acc[k] += (double)o[k];
            steps += (double)o[5]; clk += (double)o[7] / 1048576.0 * 100.0; pre += (double)o[4]; tot += (double)o[6]; ++nw;
          }
          if (nw) std::fprintf(stderr, "[k_fs_maps4_h layer %d] %.0f MHz; wave life %.0f cycles of which before the row loop %.0f; per row: loader %.0f  reads+mfma %.0f  epilogue+stores %.0f  barrier %.0f\n",
                               l, clk / nw, tot / nw, pre / nw, acc[0] / steps, acc[1] / steps, acc[2] / steps, acc[3] / steps);
        }
      } else
#endif
      {
        auto kfn = NUr == 2 ? &k_fs_maps4_h<false, 2> : &k_fs_maps4_h<false, 1>;
        const void* fn = reinterpret_cast<const void*>(kfn);
        if (ctx->lds_attr_set.insert(fn).second) SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, MH_LDS));
        // tall bands: as many bands over the stacked planes as fill the chip's workgroup slots once (bands of at least 32 rows)
        int tall_rpb = 0;
        unsigned nwg = (unsigned)(planes * hb * hs);
        static const bool tall_off = std::getenv("SS4K_MH_NO_TALL") != nullptr;
        if (!tall_off) {
          const long rows = (long)planes * h;
          const int slots = per_cu * ctx->num_cu;
          const int nb = (int)std::max(1L, std::min(rows / 32, (long)(slots / std::max(1, hs))));
          tall_rpb = (int)((rows + nb - 1) / nb);
          nwg = (unsigned)(((rows + tall_rpb - 1) / tall_rpb) * hs);
        }
        hipLaunchKernelGGL(kfn, dim3(nwg), block, MH_LDS, st, reinterpret_cast<const uint2*>(cur),
                           reinterpret_cast<uint2*>(nxt), mw, planes, h, w, hb, nullptr, tall_rpb);
      }
    } else {
      const void* fn = reinterpret_cast<const void*>(&k_fs_maps4);
      if (ctx->lds_attr_set.insert(fn).second) SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, FM_LDS));
      // tall bands: as many bands over the stacked planes as fill the chip's workgroup slots once (bands of at least 32 rows)
      static const bool tall_off = std::getenv("SS4K_MH_NO_TALL") != nullptr;
      int tall_rpb = 0;
      unsigned nwg = (unsigned)(planes * mbands * mstrips);
      if (!tall_off) {
        const long rows = (long)planes * h;
        const int nb = (int)std::max(1L, std::min(rows / 32, (long)(4 * ctx->num_cu / std::max(1, mstrips))));
        tall_rpb = (int)((rows + nb - 1) / nb);
        nwg = (unsigned)(((rows + tall_rpb - 1) / tall_rpb) * mstrips);
      }
      hipLaunchKernelGGL(k_fs_maps4, dim3(nwg), block, FM_LDS, st, cur, nxt, mw, planes, h, w, mbands, tall_rpb);
    }
    std::swap(cur, nxt);
  }
  prof_map.done(10368.0 * (double)total);
  SS4K_REQUIRE(factor == 2 || factor == 4, "FSRCNN: scale must be 2 or 4");
  ProfScope prof_tail(ctx, st, PROF_FS_TAIL);
  const int ci = exact ? FS_CI : 4 * (factor == 2 ? FsTailGeo<2>::CI : FsTailGeo<4>::CI);   // interior LR columns per workgroup
  const int strips = (w + ci - 1) / ci;
  // one round of workgroups at three per CU; every band re-does 4 halo rows (fp16 mode: 4, 5 or 6 per CU measured, no gain)
  const int bands = std::max(1, std::min((h + 15) / 16, 3 * ctx->num_cu / std::max(1, planes * strips)));
  const dim3 tgrid((unsigned)(planes * bands * strips));
  auto launch_tail = [&](auto kern, int S) {   // the exact-fp32 tail
    const size_t lds = (size_t)(6144 + 768 + 128 + 16 * (S * 128 + 8)) * 4;
    hipLaunchKernelGGL(kern, tgrid, block, lds, st, cur, out, W.w_expand, W.b_expand, W.a_expand, W.w_deconv, W.b_deconv,
                       planes, h, w, bands);
  };
  auto launch_tail_r = [&](auto kern, int S) {   // the matrix-core tails: tall bands of independent waves (see the kernel)
    const size_t lds = (size_t)(2 * 768 + 2 * 128) * 16 + 128 * 4;
    static const bool tall_off = std::getenv("SS4K_TAIL_NO_TALL") != nullptr;
    int tall_rpb = 0;
    dim3 grid = tgrid;
    if (!tall_off) {
      const int wstrips = (w + ci / 4 - 1) / (ci / 4);
      const long rows = (long)planes * h;
      const int nb = (int)std::max(1L, std::min(rows / 16, (long)(4 * 3 * ctx->num_cu / wstrips)));
      tall_rpb = (int)((rows + nb - 1) / nb);
      const long waves = ((rows + tall_rpb - 1) / tall_rpb) * wstrips;
      grid = dim3((unsigned)((waves + 3) / 4));
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, cur, out, W.w_expand, W.b_expand, W.a_expand, W.w_deconv, W.b_deconv,
                       planes, h, w, bands, tall_rpb);
  };
  if (factor == 2) {
    if (exact) launch_tail(k_fs_tail<2>, 2); else if (half && out_half) launch_tail_r(k_fs_tail_r<2, false, true>, 2);
    else if (half) launch_tail_r(k_fs_tail_r<2, false>, 2); else launch_tail_r(k_fs_tail_r<2, true>, 2);
  } else {
    if (exact) launch_tail(k_fs_tail<4>, 4); else if (half && out_half) launch_tail_r(k_fs_tail_r<4, false, true>, 4);
    else if (half) launch_tail_r(k_fs_tail_r<4, false>, 4); else launch_tail_r(k_fs_tail_r<4, true>, 4);
  }
  prof_tail.done(10416.0 * (double)total);
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
