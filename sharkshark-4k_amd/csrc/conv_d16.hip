// Dense-block layer pair on v_mfma_f32_16x16x32_f16: conv_dense.hip's fused (conv_k, conv_{k+1}) launch - shared input planes streamed
// once, x_k handed over in LDS - with conv_w16.hip's MFMA shape and phase pipeline (why that shape: conv_w16.hip; why fuse: conv_dense.hip).
//
// Geometry (fp16, 4 waves, two workgroups per CU).  A tile is 14 x 32 pixels: with the x_k ring the halo tile is 18 x 36 pixels = 20.25 KB,
// and two of them next to a ring of three 12 KB weight phases are 78.6 KB (16-row tiles: 83.2 KB - one workgroup per CU).  All LDS images of
// a tile share one coordinate system, buffer (row r, column c) <-> image (y0 - 2 + r, x0 - 2 + c), no half swizzle (conv_w16.hip):
//   input planes: rows 0..17, columns 0..35; buffer 0 holds the even plane of a chunk pair, buffer 1 the odd one;
//   x_k (j, i) for j = -1..14, i = -1..32 at buffer (j + 2, i + 2): plane 0 in buffer 0, plane 1 in buffer 1 once the input planes are done;
//   an output pixel (R, C) of either layer reads buffer (R + 1 + dy, C + 1 + dx) - conv_{k+1}'s reads of x_k use the addresses of its reads
//   of the input planes.
// Work split: wave (rgp = wave & 1, jw = wave >> 1) owns rows 7 rgp .. 7 rgp + 6 of the tile and 16-cout block jw of BOTH layers (conv_k's
// block jw = x_k's plane jw), so every wave carries the same load in every part of the tile: 56 + 56 accumulator registers, plus three
// 16-pixel groups of the ring of x_k around the tile (2 x 34 + 2 x 14 = 96 pixels = six groups per block) as per-lane gathers: 12 more.
// A chunk pair of the planes both layers read is three phases (conv_w16.hip): [dy][conv_k b0, conv_{k+1} b0, conv_k b1, conv_{k+1} b1]
// fragments = 12 KB each; conv_{k+1}'s two x_k chunks are one more chunk pair of [dy][b0, b1] = 6 KB phases (pack.cpp, pack_dense_d16).
// MFMA work against four launches: conv_k x (448 + 96) / 448 = 1.21, conv_{k+1} x 1; 360 rows = 25.7 tiles of 14 (+ 1.1 %).
// Results: not bit-identical to conv_dense.hip (another summation order inside an MFMA); same accuracy (tests/test_gpu_d16.py).
#include "common.h"
#include "conv_tile.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ss4k {
namespace d16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4, MB = 7, TH = 2 * MB;
constexpr int XH = TH + 4, XW = TW + 4;
constexpr int REC = 32;
constexpr int ROWX = XW * REC;                      // 1152
constexpr int XT_SLOTS = XH * XW * 2;               // 1296
constexpr int XT_BYTES = XT_SLOTS * 16;             // 20736
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 21
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 6
constexpr int WP = 12 * 1024, WPK = 6 * 1024, NWS = 3;
constexpr int NDMA = DMA_PER_WAVE + 3;
constexpr int W_OFF = 2 * XT_BYTES, B_OFF = W_OFF + NWS * WP;
constexpr size_t LDS_BYTES = B_OFF + 64 * 4;
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ f32x4 mma16(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}
// LeakyReLU (slope in [0, 1]) of an accumulator's four values as max(t, slope t), rounded to fp16: 8 bytes
__device__ __forceinline__ uint2 lrelu_h4(const f32x4& acc, float slope) {
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = acc[i], st = t * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(t), "v"(st));
  }
  uint2 r;
  __half* h = reinterpret_cast<__half*>(&r);
#pragma unroll
  for (int i = 0; i < 4; ++i) h[i] = __float2half(v[i]);
  return r;
}

template <bool STAMP = false>
__global__ __launch_bounds__(64 * NW, 2) void conv3x3_d16_kernel(const DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  const int rgp = wave & 1, jw = wave >> 1;
  const int K1 = a.nchunks0 + a.nchunks1, NPX = K1 >> 1;   // chunk pairs of the planes both layers read (K1 even, host-checked)
  const int NPH = 3 * NPX + 3;                             // phases per tile: the x pairs + the x_k pair
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = a.w16p;
  auto w_off = [&](int lp) __attribute__((always_inline)) -> size_t { return lp < 3 * NPX ? (size_t)lp * WP : (size_t)(3 * NPX) * WP + (size_t)(lp - 3 * NPX) * WPK; };

  // STAMP: [0] tile start, [1] x phases: reads + MFMAs + DMA issue, [2] x_k phases, [3] vmcnt wait, [4] barrier, [5] x_k epilogue, [6] its barrier,
  // [7] conv_{k+1} epilogue
  unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, rt0 = 0, ct0 = 0;
  if constexpr (STAMP) { rt0 = __builtin_amdgcn_s_memrealtime(); ct0 = __builtin_amdgcn_s_memtime(); tlast = ct0; }
  auto stamp = [&](int k) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      ph_t[k] += t - tlast;
      tlast = t;
    }
  };
  const bool banded = (gridDim.x % 8 == 0) && ntiles >= (int)gridDim.x;
  const int tpx = (ntiles + 7) / 8;
  auto tile_of = [&](int k) __attribute__((always_inline)) -> int {
    if (!banded) {
      const int t = blockIdx.x + k * gridDim.x;
      return t < ntiles ? (a.reverse ? ntiles - 1 - t : t) : -1;
    }
    const int base = (blockIdx.x & 7) * tpx, len = min(tpx, ntiles - base);
    const int j = (blockIdx.x >> 3) + k * (gridDim.x >> 3);
    return j < len ? base + (a.reverse ? len - 1 - j : j) : -1;
  };
  // operand read bases of the 7 x 32 pixels (rows 7 rgp + 1 .. + 9 of a buffer): phases 0 / 2 column 16 hn + n16 + 1 + (kg >> 1), phase 1 column
  // 16 hn + n16 + 3 of buffer kg >> 1
  int rdA[2], rdX[2];
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    rdA[hn] = (((rgp * MB + 1) * XW + 16 * hn + n16 + 1 + (kg >> 1)) * 2 + (kg & 1)) * 16;
    rdX[hn] = (((rgp * MB + 1) * XW + 16 * hn + n16 + 3) * 2 + (kg & 1)) * 16 + (kg >> 1) * XT_BYTES;
  }
  // The ring of x_k, x_k coordinates (j, i): j = -1 and j = 14 for i = -1..32, i = -1 and i = 32 for j = 0..13: 96 pixels; this wave's group g
  // takes pixel t = (3 rgp + g) * 16 + n16.  x_k (j, i) reads buffer (j + 1 + dy, i + 1 + dx) and lands at buffer (j + 2, i + 2).
  auto ring_ji = [&](int g) __attribute__((always_inline)) -> int {   // (j + 1) | (i + 1) << 8, recomputed where needed
    int n = n16;
    asm volatile("" : "+v"(n));
    const int t = (3 * rgp + g) * 16 + n;
    int j, i;
    if (t < 2 * (TW + 2)) { j = t < TW + 2 ? -1 : TH; i = (t < TW + 2 ? t : t - (TW + 2)) - 1; }
    else { const int u = t - 2 * (TW + 2); j = u < TH ? u : u - TH; i = u < TH ? -1 : TW; }
    return (j + 1) | ((i + 1) << 8);
  };
  int rdR[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) { const int ji = ring_ji(g); rdR[g] = (((ji & 0xff) * XW + (ji >> 8) + (kg >> 1)) * 2 + (kg & 1)) * 16; }

  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };
  auto tile_src = [&](int k, int n, int y0, int x0) __attribute__((always_inline)) -> uint32_t {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // computed where it is used: hoisted to the top of a phase the six offsets are registers through its MFMAs
    const int s = k * 64 + ln;
    const int p = s >> 1, gq = s & 1;
    const int row = p / XW, x = p - row * XW;
    const int iy = y0 - 2 + row, ix = x0 - 2 + x;
    const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? ((uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix) * REC + (uint32_t)(gq * 16) : OOB;
  };
  auto plane_of = [&](int c) __attribute__((always_inline)) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  const char* pf_plane = nullptr; const char* pf_w = nullptr; int pf_buf = 0, pf_slot = 0, pf_n = 0, pf_y0 = 0, pf_x0 = 0, pf_wn = 0; bool pf_tile = false;
  auto dma_op = [&](int idx) __attribute__((always_inline)) {
    if (idx < DMA_PER_WAVE) {
      const int k = wave + NW * idx;
      if (pf_tile && k < XT_DMA) {
        const uint32_t so = tile_src(k, pf_n, pf_y0, pf_x0);
        const char* src = so != OOB ? pf_plane + so : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + pf_buf * XT_BYTES + k * 1024);
        if (k * 64 + lane < XT_SLOTS) dma16(src, dst);
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - DMA_PER_WAVE);   // 1 KB piece of the requested phase's weights (12 or 6 of them)
      int l16 = lane * 16;
      asm volatile("" : "+v"(l16));   // (the 64-bit source address is formed here, not at the top of the tile for every phase at once)
      if (k < pf_wn) dma16(pf_w + k * 1024 + l16, __builtin_amdgcn_readfirstlane(lds0 + W_OFF + pf_slot * WP + k * 1024));
    }
  };
  auto slot_inc = [](int s) { return s == NWS - 1 ? 0 : s + 1; };
  // request of phase lp (0 .. NPH - 1) of the current tile: the weights of phase lp + 2 (wrapping into the next tile)
  auto w_request = [&](int lp, int ws, bool have_next) __attribute__((always_inline)) {
    int l2 = lp + 2;
    const bool wrap = l2 >= NPH;
    if (wrap) l2 -= NPH;
    pf_w = wbase + w_off(l2); pf_wn = (wrap && !have_next) ? 0 : (l2 < 3 * NPX ? 12 : 6); pf_slot = ws == 0 ? NWS - 1 : ws - 1;
  };

  float* bias_lds = reinterpret_cast<float*>(smem + B_OFF);   // [conv_k 32][conv_{k+1} 32]
  if (tid < 64) bias_lds[tid] = tid < 32 ? a.bias1[tid] : a.bias2[tid - 32];

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  pf_n = n; pf_y0 = y0; pf_x0 = x0;
  int ws = 0;
  pf_tile = true; pf_plane = plane_of(0); pf_buf = 0; pf_w = wbase; pf_wn = 12; pf_slot = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  pf_tile = false; pf_w = wbase + w_off(1); pf_slot = 1;
#pragma unroll
  for (int i = DMA_PER_WAVE; i < NDMA; ++i) dma_op(i);
  dma_wait();
  __syncthreads();
  const int lane16 = lane * 16;
  const int wsub = lane16 + jw * 2048;    // x phases: this wave's (conv_k, conv_{k+1}) blocks inside a [dy][4 blocks] table
  const int wsubk = lane16 + jw * 1024;   // x_k phases: its conv_{k+1} block inside a [dy][2 blocks] table
  uint4 wf[3][2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int e = 0; e < 2; ++e) wf[dy][e] = *reinterpret_cast<const uint4*>(smem + W_OFF + wsub + (dy * 4 + e) * 1024);
  const float slope = a.slope;

  while (true) {
    f32x4 acc1[MB][2], acc2[MB][2], accr[3];   // conv_k, conv_{k+1}: [row][pixel half]; conv_k on the ring groups
    f32x4 bias4[2];
    {
      const float4* bp = reinterpret_cast<const float4*>(smem + B_OFF + (jw * 16 + kg * 4) * 4);
      int o = 0;
      asm volatile("" : "+v"(o));
      const float4 b0 = bp[o], b1 = bp[o + 8];
      bias4[0] = f32x4{b0.x, b0.y, b0.z, b0.w}; bias4[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    stamp(0);

    // one phase.  XK: the x_k chunk pair (conv_{k+1} only, 6 KB weight phases, no ring); lp: the phase's index in the tile (runtime)
    auto phase = [&](auto PH, auto FIRST, auto XKT, const int lp, const int q) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value;
      constexpr bool first = decltype(FIRST)::value;
      constexpr bool XK = decltype(XKT)::value;
      w_request(lp, ws, next_tile >= 0);
      if constexpr (!XK) {
        if constexpr (ph == 0) { pf_tile = true; pf_plane = plane_of(2 * q + 1); pf_buf = 1; }
        else if constexpr (ph == 1) { pf_tile = false; }
        else { pf_buf = 0; pf_tile = q + 1 < NPX; if (pf_tile) pf_plane = plane_of(2 * q + 2); }
      } else {
        if constexpr (ph == 2) {
          pf_buf = 0;
          if (next_tile >= 0) { setup_tile(next_tile, n, y0, x0); pf_n = n; pf_y0 = y0; pf_x0 = x0; pf_tile = true; pf_plane = plane_of(0); }
          else pf_tile = false;
        } else pf_tile = false;
      }
      const bool more = lp + 1 < NPH || next_tile >= 0;
      const bool next_xk = lp + 1 >= 3 * NPX && lp + 1 < NPH;   // the next phase reads a [dy][2 blocks] table
      const char* tb = smem + (ph == 2 ? XT_BYTES : 0);
      const int* rd = ph == 1 ? rdX : rdA;
      const char* wbn = smem + W_OFF + slot_inc(ws) * WP;
      uint4 bf[3][2], rf;
      auto bf_load = [&](int t, int hn) __attribute__((always_inline)) { return *reinterpret_cast<const uint4*>(tb + rd[hn] + t * ROWX); };
      // ring step s = 0..8: group s / 3, tap row s % 3
      auto rf_load = [&](int s) __attribute__((always_inline)) {
        const int base = rdR[s / 3] + (ph == 1 ? 64 + (kg >> 1) * (XT_BYTES - 32) : 0);
        return *reinterpret_cast<const uint4*>(tb + base + (s % 3) * ROWX);
      };
#pragma unroll
      for (int t = 0; t < 3; ++t) { bf[t][0] = bf_load(t, 0); bf[t][1] = bf_load(t, 1); }
      if constexpr (!XK) rf = rf_load(0);
      int m = 0;
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ir = 0; ir < MB + 2; ++ir) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int r = ir - dy;
          if (r >= 0 && r < MB) {
#pragma unroll
            for (int hn = 0; hn < 2; ++hn) {
              if constexpr (!XK) acc1[r][hn] = mma16(wf[dy][0], bf[ir % 3][hn], (first && dy == 0) ? bias4[0] : acc1[r][hn]);
              acc2[r][hn] = mma16(wf[dy][1], bf[ir % 3][hn], (first && dy == 0) ? bias4[1] : acc2[r][hn]);
              // DMA slots (42 MFMA groups per phase): the halo tile after groups 1, 3, .. 11, the weights after groups 15, 19, 23
              const int sl = (m & 1) && m < 12 ? m >> 1 : (m == 15 || m == 19 || m == 23) ? DMA_PER_WAVE + (m - 15) / 4 : -1;
              if (sl >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(sl);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
        }
        if constexpr (!XK) {   // ring step ir: one MFMA of conv_k on a gathered group
          accr[ir / 3] = mma16(wf[ir % 3][0], rf, (first && ir % 3 == 0) ? bias4[0] : accr[ir / 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!XK) { if (ir + 1 < MB + 2) rf = rf_load(ir + 1); }
        if (ir + 3 < MB + 2) { bf[ir % 3][0] = bf_load(ir + 3, 0); bf[ir % 3][1] = bf_load(ir + 3, 1); }
        if (more && ir >= MB - 1) {   // dy = ir - 6 has issued its last MFMA of this phase: its fragments of the next one
          const int dy = ir - (MB - 1);
          if (next_xk) wf[dy][1] = *reinterpret_cast<const uint4*>(wbn + wsubk + dy * 2048);
          else {
            wf[dy][0] = *reinterpret_cast<const uint4*>(wbn + wsub + (dy * 4) * 1024);
            wf[dy][1] = *reinterpret_cast<const uint4*>(wbn + wsub + (dy * 4 + 1) * 1024);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      stamp(XK ? 2 : 1);
      if (more) { dma_wait(); stamp(3); __syncthreads(); stamp(4); }
      ws = slot_inc(ws);
    };
    auto period = [&](const int q, auto FIRSTP) __attribute__((always_inline)) {
      phase(std::integral_constant<int, 0>{}, FIRSTP, std::false_type{}, 3 * q, q);
      phase(std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{}, 3 * q + 1, q);
      phase(std::integral_constant<int, 2>{}, std::false_type{}, std::false_type{}, 3 * q + 2, q);
    };
    period(0, std::true_type{});
#pragma unroll 1
    for (int q = 1; q < NPX; ++q) period(q, std::false_type{});
    // here: every wave is past the input planes (the barrier that ended the last phase); both buffers are free

    // ---------------- x_k: LeakyReLU, fp16, -> LDS at buffer (j + 2, i + 2) of buffer jw (zeros outside the image: conv_{k+1}'s padding) and
    // -> memory (the tile's own pixels); lane (pixel, kg) holds channels 4 kg .. 4 kg + 3 of plane jw: 8 bytes
    {
      int kge = kg, ne = n16;
      asm volatile("" : "+v"(kge), "+v"(ne));   // addresses are formed here, per tile (hoisted out of the tile loop they are registers of the MFMA loops)
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      char* o1 = a.out1 + (size_t)(a.out1_plane0 + jw) * a.out1_plane_bytes + (size_t)kge * 8;
      // the tile's own pixels: x_k (j, i) = (7 rgp + r, 16 hn + n16), never left of / above the image
      {
        char* xl = smem + jw * XT_BYTES + ((rgp * MB + 2) * XW + ne + 2) * REC + kge * 8;
        const int yb = cur_y0 + rgp * MB, xc = cur_x0 + ne;
        char* og = o1 + ((size_t)(cur_n * a.H + yb) * a.W + xc) * REC;
#pragma unroll
        for (int r = 0; r < MB; ++r)
#pragma unroll
          for (int hn = 0; hn < 2; ++hn) {
            uint2 h = lrelu_h4(acc1[r][hn], slope);
            const bool in = yb + r < a.H && xc + 16 * hn < a.W;
            const uint32_t msk = in ? 0xFFFFFFFFu : 0u;
            *reinterpret_cast<uint2*>(xl + (r * XW + 16 * hn) * REC) = make_uint2(h.x & msk, h.y & msk);
            if (in) __builtin_nontemporal_store(*reinterpret_cast<u32x2*>(&h), reinterpret_cast<u32x2*>(og + ((size_t)r * a.W + 16 * hn) * REC));
          }
      }
      // the ring: LDS only
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int ji = ring_ji(g), j = (ji & 0xff) - 1, i = (ji >> 8) - 1;
        const int y = cur_y0 + j, x = cur_x0 + i;
        uint2 h = lrelu_h4(accr[g], slope);
        const uint32_t msk = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? 0xFFFFFFFFu : 0u;
        *reinterpret_cast<uint2*>(smem + jw * XT_BYTES + ((j + 2) * XW + i + 2) * REC + kge * 8) = make_uint2(h.x & msk, h.y & msk);
      }
    }
    stamp(5);
    lds_barrier();
    stamp(6);
    // ---------------- conv_{k+1}'s last two K-chunks: x_k from LDS, one more chunk pair
    phase(std::integral_constant<int, 0>{}, std::false_type{}, std::true_type{}, 3 * NPX, 0);
    phase(std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{}, 3 * NPX + 1, 0);
    phase(std::integral_constant<int, 2>{}, std::false_type{}, std::true_type{}, 3 * NPX + 2, 0);

    // ---------------- conv_{k+1}'s epilogue
    {
      int kge = kg, ne = n16;
      asm volatile("" : "+v"(kge), "+v"(ne));
      typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
      const int yb = cur_y0 + rgp * MB, xc = cur_x0 + ne;
      char* og = a.out2 + (size_t)(a.out2_plane0 + jw) * a.out2_plane_bytes + (size_t)kge * 8 + ((size_t)(cur_n * a.H + yb) * a.W + xc) * REC;
#pragma unroll
      for (int r = 0; r < MB; ++r)
#pragma unroll
        for (int hn = 0; hn < 2; ++hn) {
          uint2 h = lrelu_h4(acc2[r][hn], slope);
          if (yb + r < a.H && xc + 16 * hn < a.W)
            __builtin_nontemporal_store(*reinterpret_cast<u32x2*>(&h), reinterpret_cast<u32x2*>(og + ((size_t)r * a.W + 16 * hn) * REC));
        }
    }
    stamp(7);
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg_buf && blockIdx.x < 1024) {
      unsigned long long* o = a.dbg_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
      for (int k = 0; k < 8; ++k) o[k] = ph_t[k];
      o[11] = (unsigned long long)(kt + 1);
      o[12] = ((__builtin_amdgcn_s_memtime() - ct0) << 20) / (__builtin_amdgcn_s_memrealtime() - rt0 + 1);
    }
  }
}

}  // namespace d16

bool conv3x3_d16_eligible(int nchunks_a, int cout_pad_a, int nchunks_b, int cout_pad_b) {
  return cout_pad_a == 32 && cout_pad_b == 32 && nchunks_b == nchunks_a + 2 && nchunks_a >= 2 && nchunks_a % 2 == 0;
}

void launch_conv3x3_d16(ss4k_ctx* ctx, const DenseArgs& a0, hipStream_t st) {
  using namespace d16;
  DenseArgs a = a0;
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.w16p, "dense pair (d16): empty grid or no weights");
  SS4K_REQUIRE((a.nchunks0 + a.nchunks1) % 2 == 0 && a.nchunks0 + a.nchunks1 >= 2, "dense pair: conv_k needs an even number of K-chunks");
  SS4K_REQUIRE(a.slope >= 0.f && a.slope <= 1.f, "dense pair: LeakyReLU slope must be in [0,1]");
  SS4K_REQUIRE((double)(a.n0 + a.N) * a.H * a.W * 32.0 < 4294967296.0, "dense pair: a plane holds at most 4 GB");
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  ProfScope prof(ctx, st, PROF_CONV);
#ifdef SS4K_DEV
  static const bool stamp_mode = std::getenv("SS4K_D16_STAMP") && std::getenv("SS4K_D16_STAMP")[0] == '1';
  if (stamp_mode) {
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) SS4K_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 4 * 16 * 8));
    SS4K_HIP(hipMemsetAsync(dbuf, 0, 1024 * 4 * 16 * 8, st));
    a.dbg_buf = dbuf;
    const void* fs = reinterpret_cast<const void*>(&conv3x3_d16_kernel<true>);
    if (ctx->lds_attr_set.insert(fs).second) SS4K_HIP(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(conv3x3_d16_kernel<true>, dim3(gx), dim3(64 * NW), LDS_BYTES, st, a);
    SS4K_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hb(1024 * 4 * 16);
    SS4K_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
    static int printed = 0;
    if (printed++ < 6) {
      double acc[8] = {0}, tiles = 0, clk = 0; int nw = 0;
      for (size_t i = 0; i < 1024 * 4; ++i) {
        const unsigned long long* o = &hb[i * 16];
        if (!o[11]) continue;
        for (int k = 0; k < 8; ++k) acc[k] += (double)o[k];
        tiles += (double)o[11]; clk += (double)o[12] / 1048576.0 * 100.0; ++nw;
      }
      if (tiles > 0) {
        double tot = 0; for (double v : acc) tot += v;
        std::fprintf(stderr, "[d16 K1=%d] %.0f MHz, cycles per tile and wave: total %.0f | start %.0f | x phases %.0f | x_k phases %.0f | vmcnt wait %.0f | barrier %.0f | x_k epilogue %.0f + barrier %.0f | epilogue %.0f\n",
                     a.nchunks0 + a.nchunks1, clk / nw, tot / tiles, acc[0] / tiles, acc[1] / tiles, acc[2] / tiles, acc[3] / tiles, acc[4] / tiles, acc[5] / tiles, acc[6] / tiles, acc[7] / tiles);
      }
    }
    prof.done(a0.flops);
    return;
  }
#endif
  const void* fn = reinterpret_cast<const void*>(&conv3x3_d16_kernel<false>);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  hipLaunchKernelGGL(conv3x3_d16_kernel<false>, dim3(gx), dim3(64 * NW), LDS_BYTES, st, a);
  SS4K_HIP(hipGetLastError());
  prof.done(a0.flops);
}

}  // namespace ss4k
