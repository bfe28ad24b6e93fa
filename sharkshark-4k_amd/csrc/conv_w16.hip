// conv3x3_w16_kernel: ONE 3x3 layer with 64 output channels per workgroup - conv_dense.hip's single-layer kernel (same tile, same LDS
// budget, same DMA pipeline, two workgroups per CU) on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
//
// Why.  The part runs these layers at its 1400 W cap (profiles/earlier/r04/r04_headline_power_clock.txt: 1.82 of 2.4 GHz through the whole timed
// loop), and at equal cycles per FLOP the 16x16x32 shape costs less energy: 1.12-1.14 x the FLOP/s under the cap with operands
// re-read from LDS at this loop's reuse ratios (tools/micro/mfma_shapes.hip).  conv_rs.hip has used the shape since round 2, paying for
// its K = 32 with 64-byte-per-pixel halo stages (one tap x TWO planes per MFMA) and one workgroup per CU.  Here K = 32 is
//   * two horizontally adjacent taps of ONE plane - (dx 0, dx 1) x 16 channels: the operand of k-group kg = lane >> 4 is the 16-byte
//     half (kg & 1) of pixel column + (kg >> 1), so the planes stay 32-byte records in LDS and a halo tile stays 19 KB;
//   * and, for the third tap column, dx 2 of TWO planes: k-groups 2, 3 read the same pixel of the other tile buffer.
// A pair of K-chunks (planes 2q, 2q + 1; buffer 0 always holds the even plane, buffer 1 the odd one) is three PHASES:
//   phase 0: plane 2q, taps (dx 0, dx 1)      phase 1: planes 2q and 2q + 1, tap dx 2      phase 2: plane 2q + 1, taps (dx 0, dx 1)
// each 3 (dy) x 4 (16-cout blocks) weight fragments = 12 KB, 96 MFMAs per wave, no padded K anywhere (conv_rs.hip's intra-plane
// pairing of nine taps would waste 1 in 10).  A wave owns 8 rows x 32 pixels x 32 couts (waves 0, 1: rows 0-7 / 8-15 of the group's first
// two 16-cout blocks, waves 2, 3: of the other two): 128 accumulator registers, six weight fragments live (twelve - every wave all 64
// couts on 4 rows - spilled 69 registers), an activation fragment (16 pixels x K 32) feeds 3 dy x 2 blocks = 6 MFMAs of 16 cycles, a
// weight fragment 8 rows x 2 pixel halves = 16: 26 LDS reads per 96 MFMAs against the 32x32x16 loop's 12 per 24 of twice the size.
// Pipeline: buffer 1 (plane 2q + 1) is requested during phase 0 and awaited at its end, buffer 0 (plane 2q + 2) during phase 2; the
// weights of phase g + 2 are requested during phase g into the slot phase g - 1 has just left (a ring of three 12 KB slots); one
// s_waitcnt vmcnt(0) + barrier per phase.  Weight fragments are re-filled in place for the next phase as soon as their last MFMA of
// this one has issued.  LDS images carry NO half swizzle: a ds_read_b128 pass serves lanes {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
// of a half-wave, i.e. sixteen consecutive pixels with halves (0 x 4, 1 x 8, 0 x 4) - distinct bank quads exactly when pixel c and
// pixel c + 8 sit in the same half order.
// Results: NOT bit-identical to the 32x32x16 kernels (an MFMA sums 32 products of two taps where the other sums 16 of one); within
// fp32 accumulation noise of them (tests/test_gpu_w16.py) - the same relation conv_rs.hip's results have to conv_mfma.hip's.
#include "common.h"
#include <cmath>
#include "conv_tile.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ss4k {
namespace w16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4, MB = 8, TH = 16;   // waves, rows per wave (two row groups x two cout halves), tile rows
constexpr int XH = TH + 2, XW = TW + 2;
constexpr int REC = 32;
constexpr int ROWX = XW * REC;                      // 1088
constexpr int XT_SLOTS = XH * XW * 2;               // 1224
constexpr int XT_BYTES = XT_SLOTS * 16;             // 19584
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 20
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 5
constexpr int WP = 12 * 1024, NWS = 3;              // a phase's weights [dy][cout block][lane], ring slots
constexpr int NDMA = DMA_PER_WAVE + 3;              // per wave and phase: 5 halo-tile pieces, 3 of the 12 weight pieces
constexpr int W_OFF = 2 * XT_BYTES, B_OFF = W_OFF + NWS * WP;
constexpr size_t LDS_BYTES = B_OFF + 2 * 64 * 4;    // + bias (accumulator order) and slope (virtual cout order) of the group's 64 couts
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ f32x4 mma16(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}

//   STAMP (dev library, SS4K_W16_STAMP=1): per-wave cycle totals of the tile's parts (s_memtime), see launch_conv3x3_w16
//   RL: res1 is the layer's own input tensor and its first four K-chunks are that tensor's planes (conv5 of an RDB: out = conv * alpha + x,
//   no activation): x's centre pixels pass through LDS as the dx 1 half of phases 0 / 2 of the first two chunk pairs, so they are added
//   there - one more MFMA per accumulator with a (1 / alpha) I fragment on tap (dy 1, dx 1), as conv_rs.hip and the wide kernel do -
//   instead of being read from memory again in the epilogue.  Output plane P = 2 jw + (kg >> 1) is input plane P: wave pair jw adds during
//   chunk pair q == jw, rows of row groups kg >> 1 == 0 in phase 0 (plane 2 jw), kg >> 1 == 1 in phase 2 (plane 2 jw + 1).
template <bool STAMP = false, bool RL = false>
__global__ __launch_bounds__(64 * NW, 2) void conv3x3_w16_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;   // operand: pixel / k-group; result: pixel / row group (couts 4 kg .. 4 kg + 3 of a block)
  const int rgp = wave & 1, jw = wave >> 1;    // this wave: tile rows 8 rgp .. 8 rgp + 7, 16-cout blocks 2 jw and 2 jw + 1
  const int grp = blockIdx.y;
  const int K = a.nchunks0 + a.nchunks1, NP = K >> 1;   // K is even (host-checked)
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = reinterpret_cast<const char*>(a.w16) + (size_t)grp * NP * (3 * WP);

  // STAMP: [0] accumulator init, [1] phases 0 and 2: reads + MFMAs + DMA issue, [2] phase 1 the same, [3] vmcnt wait, [4] barrier, [5] epilogue
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0, rt0 = 0, ct0 = 0;
  if constexpr (STAMP) { rt0 = __builtin_amdgcn_s_memrealtime(); ct0 = __builtin_amdgcn_s_memtime(); tlast = ct0; }
  auto stamp = [&](int k) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      ph[k] += t - tlast;
      tlast = t;
    }
  };
  const bool banded = (gridDim.x % 8 == 0) && ntiles >= (int)gridDim.x && !a.no_band;
  const int tpx = (ntiles + 7) / 8;
  auto tile_of = [&](int k) -> int {
    if (!banded) {
      const int t = blockIdx.x + k * gridDim.x;
      return t < ntiles ? (a.reverse ? ntiles - 1 - t : t) : -1;
    }
    const int base = (blockIdx.x & 7) * tpx, len = min(tpx, ntiles - base);
    const int j = (blockIdx.x >> 3) + k * (gridDim.x >> 3);
    return j < len ? base + (a.reverse ? len - 1 - j : j) : -1;
  };
  // operand read bases, buffer (row r, column c) <-> image (y0 - 1 + r, x0 - 1 + c); row group rgp reads rows 8 rgp .. 8 rgp + 9.
  //   phases 0 / 2 (one plane, taps dx 0 | dx 1): column 16 hn + n16 + (kg >> 1); phase 1 (dx 2 of both planes): column 16 hn + n16 + 2 of
  //   buffer kg >> 1
  int rdA[2], rdX[2];
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    rdA[hn] = (((rgp * MB) * XW + 16 * hn + n16 + (kg >> 1)) * 2 + (kg & 1)) * 16;
    rdX[hn] = (((rgp * MB) * XW + 16 * hn + n16 + 2) * 2 + (kg & 1)) * 16 + (kg >> 1) * XT_BYTES;
  }

  // (n, y0, x0): the tile whose planes are being REQUESTED (the next tile's from the last phase of the current one on)
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };
  // per-lane source of halo-tile DMA instruction k: LDS slot s = 64 k + lane holds pixel (row, x) = (s / 2) divmod 34, half s & 1; byte offset
  // inside a plane, OOB = the zero page.  Recomputed at every issue (a division by a constant) instead of five registers held through the MFMA loops.
  auto tile_src = [&](int k, int n, int y0, int x0) -> uint32_t {
    const int s = k * 64 + lane;
    const int p = s >> 1, gq = s & 1;
    const int row = p / XW, x = p - row * XW;
    const int iy = y0 - 1 + row, ix = x0 - 1 + x;
    const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? ((uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix) * REC + (uint32_t)(gq * 16) : OOB;
  };
  auto plane_of = [&](int c) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  // prefetch of the running phase: an optional halo tile into buffer pf_buf and an optional 12 KB of weights into ring slot pf_slot
  const char* pf_plane = nullptr; const char* pf_w = nullptr; int pf_buf = 0, pf_slot = 0, pf_n = 0, pf_y0 = 0, pf_x0 = 0; bool pf_tile = false, pf_wt = false;
  auto dma_op = [&](int idx) {
    if (idx < DMA_PER_WAVE) {
      const int k = wave + NW * idx;
      if (pf_tile && k < XT_DMA) {
        const uint32_t so = tile_src(k, pf_n, pf_y0, pf_x0);
        const char* src = so != OOB ? pf_plane + so : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + pf_buf * XT_BYTES + k * 1024);
        if (k * 64 + lane < XT_SLOTS) dma16(src, dst);
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - DMA_PER_WAVE);   // piece 0..11 of the phase's weights
      if (pf_wt) dma16(pf_w + k * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + W_OFF + pf_slot * WP + k * 1024));
    }
  };
  auto slot_inc = [](int s) { return s == NWS - 1 ? 0 : s + 1; };

  float* epi_lds = reinterpret_cast<float*>(smem + B_OFF);   // [64 bias, accumulator order (block, row group, i)][64 slope, virtual cout order]
  if (tid < 64) {
    const int mbk = tid >> 4, q = (tid >> 2) & 3, i = tid & 3;
    const int vb = grp * 64 + 32 * (mbk >> 1) + 16 * (q >> 1) + 8 * (q & 1) + 4 * (mbk & 1) + i;   // pack.cpp, pack_conv3x3_w16
    epi_lds[tid] = vb < a.cout_pad ? a.bias[vb] : 0.f;
    const int v = grp * 64 + tid;
    epi_lds[64 + tid] = a.act == ACT_PRELU ? (v < a.cout_pad ? a.prelu[v] : 1.f) : (a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU6 ? 0.f : 1.f));
  }

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  pf_n = n; pf_y0 = y0; pf_x0 = x0;
  int ws = 0;   // ring slot of the running phase's weights
  // before phase 0 of the first tile: buffer 0 <- plane 0, weights of phases 0 and 1
  pf_tile = true; pf_plane = plane_of(0); pf_buf = 0; pf_wt = true; pf_w = wbase; pf_slot = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  pf_tile = false; pf_w = wbase + WP; pf_slot = 1;
#pragma unroll
  for (int i = DMA_PER_WAVE; i < NDMA; ++i) dma_op(i);
  dma_wait();
  __syncthreads();
  const int lane16 = lane * 16;
  const int wsub = lane16 + jw * 2048;   // this wave's two blocks inside a [dy][block] fragment table
  uint4 wf[3][2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int e = 0; e < 2; ++e) wf[dy][e] = *reinterpret_cast<const uint4*>(smem + W_OFF + wsub + (dy * 4 + e) * 1024);

  while (true) {
    f32x4 acc[MB][2][2];   // [output row][pixel half][block 2 jw + e]; written whole by its first MFMA of the tile (C = the bias, as conv_rs.hip does)
    f32x4 bias4[2];
    {
      const float4* bp = reinterpret_cast<const float4*>(smem + B_OFF + kg * 16 + jw * 128);
      int o = 0;
      asm volatile("" : "+v"(o));   // re-read per tile: held across tiles the eight values are registers of the MFMA loops
      const float4 b0 = bp[o], b1 = bp[o + 4];
      bias4[0] = f32x4{b0.x, b0.y, b0.z, b0.w}; bias4[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    stamp(0);

    auto period = [&](const int q, auto FIRSTP) {
      const bool last_q = q + 1 == NP;
      auto phase = [&](auto PH, auto FIRST) {
        constexpr int ph = decltype(PH)::value;
        constexpr bool first = decltype(FIRST)::value;   // the tile's first phase: dy = 0 is an accumulator's first MFMA
        // what this phase requests.  Weights: those of phase lp + 2 (lp = 3 q + ph), wrapping into the next tile's first phases.
        const int lp2 = 3 * q + ph + 2;
        if (lp2 < 3 * NP) { pf_wt = true; pf_w = wbase + (size_t)lp2 * WP; }
        else if (next_tile >= 0) { pf_wt = true; pf_w = wbase + (size_t)(lp2 - 3 * NP) * WP; }
        else pf_wt = false;
        pf_slot = ws == 0 ? NWS - 1 : ws - 1;   // the slot phase g - 1 used = slot of phase g + 2
        // Tiles: phase 0 -> buffer 1 <- plane 2q + 1; phase 2 -> buffer 0 <- plane 2q + 2 (or the next tile's plane 0)
        if constexpr (ph == 0) { pf_tile = true; pf_plane = plane_of(2 * q + 1); pf_buf = 1; }
        else if constexpr (ph == 1) { pf_tile = false; }
        else {
          pf_buf = 0;
          if (!last_q) { pf_tile = true; pf_plane = plane_of(2 * q + 2); }
          else if (next_tile >= 0) { setup_tile(next_tile, n, y0, x0); pf_n = n; pf_y0 = y0; pf_x0 = x0; pf_tile = true; pf_plane = plane_of(0); }
          else pf_tile = false;
        }
        const bool more = !(last_q && ph == 2) || next_tile >= 0;   // a phase follows: re-fill the weight fragments for it
        const char* tb = smem + (ph == 2 ? XT_BYTES : 0);
        const int* rd = ph == 1 ? rdX : rdA;
        const char* wbn = smem + W_OFF + slot_inc(ws) * WP + wsub;
        uint4 bf[3][2];
        auto bf_load = [&](int t, int hn) { return *reinterpret_cast<const uint4*>(tb + rd[hn] + t * ROWX); };
        // RL: A fragments of (1 / alpha) I for this phase's plane.  As an A operand this lane is MFMA row m = lane & 15 (row group m >> 2,
        // i = m & 3: channel 8 (rg & 1) + 4 e2 + i of plane 2 jw + (rg >> 1) in block e2) and k-group kg: elements e = channel 8 (kg & 1) + e
        // of tap dx = kg >> 1 - the one at e = 4 e2 + i meets its row when kg >= 2 (dx 1), kg & 1 == rg & 1 and rg >> 1 == the plane's parity
        bool rl_on = false;
        uint32_t idw[2] = {0u, 0u};   // the two words of the lane's four elements 4 e2 .. 4 e2 + 3 (the other six words of a fragment are zero)
        if constexpr (RL && ph != 1) {
          rl_on = q == jw;
          const int rg = (lane & 15) >> 2, i = lane & 3;
          const bool hit = kg >= 2 && (kg & 1) == (rg & 1) && (rg >> 1) == (ph >> 1);
          const uint32_t hv = hit ? (uint32_t)__half_as_ushort(__float2half(1.f / a.alpha)) << (16 * (i & 1)) : 0u;
          idw[0] = (i >> 1) == 0 ? hv : 0u; idw[1] = (i >> 1) == 1 ? hv : 0u;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { bf[t][0] = bf_load(t, 0); bf[t][1] = bf_load(t, 1); }
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
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[r][hn][e] = mma16(wf[dy][e], bf[ir % 3][hn], (first && dy == 0) ? bias4[e] : acc[r][hn][e]);
                if constexpr (RL && ph != 1) {
                  if (dy == 1 && rl_on) {   // wave-uniform
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                      acc[r][hn][e] = mma16(e == 0 ? make_uint4(idw[0], idw[1], 0u, 0u) : make_uint4(0u, 0u, idw[0], idw[1]), bf[ir % 3][hn], acc[r][hn][e]);
                  }
                }
                // DMA slots (48 MFMA pairs per phase): the halo tile after pairs 1, 3, 5, 7, 9, the weights after pairs 13, 17, 21
                const int sl = (m & 1) && m < 10 ? m >> 1 : (m == 13 || m == 17 || m == 21) ? DMA_PER_WAVE + (m - 13) / 4 : -1;
                if (sl >= 0) {
                  __builtin_amdgcn_sched_barrier(0);
                  dma_op(sl);
                  __builtin_amdgcn_sched_barrier(0);
                }
                ++m;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ir + 3 < MB + 2) { bf[ir % 3][0] = bf_load(ir + 3, 0); bf[ir % 3][1] = bf_load(ir + 3, 1); }
          if (more && ir >= MB - 1) {   // dy = ir - 7 has issued its last MFMA of this phase
            const int dy = ir - (MB - 1);
#pragma unroll
            for (int e = 0; e < 2; ++e) wf[dy][e] = *reinterpret_cast<const uint4*>(wbn + (dy * 4 + e) * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        stamp(ph == 1 ? 2 : 1);
        if (more) { dma_wait(); stamp(3); __syncthreads(); stamp(4); }
        ws = slot_inc(ws);
      };
      phase(std::integral_constant<int, 0>{}, FIRSTP);
      phase(std::integral_constant<int, 1>{}, std::false_type{});
      phase(std::integral_constant<int, 2>{}, std::false_type{});
    };
    period(0, std::true_type{});   // (peeled: the accumulators do not exist before it)
#pragma unroll 1
    for (int q = 1; q < NP; ++q) period(q, std::false_type{});

    // ---------------- epilogue: lane (pixel n16 of half hn, row group kg) holds channels 8 (kg & 1) .. + 7 of plane 2 jw + (kg >> 1) of the
    // group's four output planes in its two blocks (first four | last four): one 16-byte store per row and pixel half
    {
      const float alpha = a.alpha, gamma = a.gamma;
      int kge = kg;
      asm volatile("" : "+v"(kge));
      float slope_v[8];
      {
        const float4* sp = reinterpret_cast<const float4*>(epi_lds + 64 + 32 * jw + 8 * kge);
        const float4 s0 = sp[0], s1 = sp[1];
        slope_v[0] = s0.x; slope_v[1] = s0.y; slope_v[2] = s0.z; slope_v[3] = s0.w;
        slope_v[4] = s1.x; slope_v[5] = s1.y; slope_v[6] = s1.z; slope_v[7] = s1.w;
      }
      const int opl = grp * 4 + 2 * jw + (kge >> 1);
      const size_t sub = (size_t)(kge & 1) * 16;
      const char* r1p = (a.res1 && !RL) ? a.res1 + (size_t)(a.r1_plane0 + opl) * a.r1_plane_bytes + sub : nullptr;
      const char* r2p = a.res2 ? a.res2 + (size_t)(a.r2_plane0 + opl) * a.r2_plane_bytes + sub : nullptr;
      char* outp = a.out + (size_t)(a.out_plane0 + opl) * a.out_plane_bytes + sub;
      const bool resid = r1p || r2p;
      // one copy of the store loop per epilogue form (a single loop with the forms selected per value compiled to eleven vector
      // instructions per value: every form evaluated, the result picked with v_cndmask)
      //   0: t >= 0 ? t : t * slope (PReLU per channel)   1: max(t, t * slope) (LeakyReLU, slope in [0, 1]; none: slope 1)   2: ReLU6
      auto stores = [&](auto FORM, auto ALPHA1) {
        constexpr int form = decltype(FORM)::value;
        constexpr bool alpha1 = decltype(ALPHA1)::value;   // alpha == 1: no multiply (x * 1.0f is x)
#pragma unroll
        for (int r = 0; r < MB; ++r) {
          const int y = cur_y0 + rgp * MB + r;
#pragma unroll
          for (int hn = 0; hn < 2; ++hn) {
            const int x = cur_x0 + 16 * hn + n16;
            const bool ok = y < a.H && x < a.W;
            const size_t rec = (((size_t)cur_n * a.H + y) * a.W + x) * REC;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float t = i < 4 ? acc[r][hn][0][i] : acc[r][hn][1][i - 4];
              float u;
              if constexpr (form == 0) { const float neg = t * slope_v[i]; u = t >= 0.f ? t : neg; }
              else if constexpr (form == 1) {
                const float st = t * slope_v[i];
                asm("v_max_f32 %0, %1, %2" : "=v"(u) : "v"(t), "v"(st));   // = fmaxf(t, st) for every non-NaN input, one instruction
              } else u = __builtin_amdgcn_fmed3f(t, 0.f, 6.f);
              v[i] = alpha1 ? u : u * alpha;
            }
            if (ok) store8<__half>(outp + rec, v);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // Residual epilogues (any activation, then * alpha + res1, * gamma + res2; absent residuals add nothing).  The residual records of
      // FOUR rows (two with both residuals) are requested together, unconditionally (coordinates clamped into the image: a load under a branch makes hipcc wait with
      // vmcnt(0) right behind it - sixteen exposed round trips per tile, + 12 % on conv5 of every third RDB), then the four rows are stored.
      auto stores_res = [&](auto R1T, auto R2T) {
        constexpr bool R1 = decltype(R1T)::value, R2 = decltype(R2T)::value;
        constexpr int RB = (R1 && R2) ? 2 : 4;   // rows per request: 8 registers per row and residual next to 128 accumulators
#pragma unroll
        for (int r0 = 0; r0 < MB; r0 += RB) {
          uint4 q1[RB][2], q2[RB][2];
#pragma unroll
          for (int rr = 0; rr < RB; ++rr)
#pragma unroll
            for (int hn = 0; hn < 2; ++hn) {
              const int yc = min(cur_y0 + rgp * MB + r0 + rr, a.H - 1), xc = min(cur_x0 + 16 * hn + n16, a.W - 1);
              const size_t recc = (((size_t)cur_n * a.H + yc) * a.W + xc) * REC;
              if constexpr (R1) q1[rr][hn] = *reinterpret_cast<const uint4*>(r1p + recc);
              if constexpr (R2) q2[rr][hn] = *reinterpret_cast<const uint4*>(r2p + recc);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr, y = cur_y0 + rgp * MB + r;
#pragma unroll
            for (int hn = 0; hn < 2; ++hn) {
              const int x = cur_x0 + 16 * hn + n16;
              const bool ok = y < a.H && x < a.W;
              const size_t rec = (((size_t)cur_n * a.H + y) * a.W + x) * REC;
              float r1[8], r2[8], v[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) { r1[i] = 0.f; r2[i] = 0.f; }
              if constexpr (R1) load8<__half>(reinterpret_cast<const char*>(&q1[rr][hn]), r1);
              if constexpr (R2) load8<__half>(reinterpret_cast<const char*>(&q2[rr][hn]), r2);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                float t = i < 4 ? acc[r][hn][0][i] : acc[r][hn][1][i - 4];
                const float neg = t * slope_v[i];
                t = t >= 0.f ? t : neg;
                if (a.act == ACT_RELU6) t = fminf(t, 6.f);
                t = t * alpha + r1[i];
                v[i] = t * gamma + r2[i];   // (conv_dense.hip's wide kernel, the same expressions)
              }
              if (ok) store8<__half>(outp + rec, v);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // PReLU with every slope of the layer <= 1 (host-checked, ConvArgs.prelu_le1): t >= 0 ? t : t s  ==  max(t, t s), one instruction less per value
      const bool max_form = a.act != ACT_PRELU || a.prelu_le1;
      const bool a1 = alpha == 1.f;
      if (r1p && r2p) stores_res(std::true_type{}, std::true_type{});
      else if (r1p) stores_res(std::true_type{}, std::false_type{});
      else if (r2p) stores_res(std::false_type{}, std::true_type{});
      else if (a.act == ACT_RELU6) stores(std::integral_constant<int, 2>{}, std::false_type{});
      else if (max_form) { if (a1) stores(std::integral_constant<int, 1>{}, std::true_type{}); else stores(std::integral_constant<int, 1>{}, std::false_type{}); }
      else { if (a1) stores(std::integral_constant<int, 0>{}, std::true_type{}); else stores(std::integral_constant<int, 0>{}, std::false_type{}); }
    }
    stamp(5);
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg_buf && blockIdx.x < 1024 && blockIdx.y == 0) {
      unsigned long long* o = a.dbg_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
      for (int k = 0; k < 6; ++k) o[k] = ph[k];
      o[6] = (unsigned long long)(kt + 1);
      o[7] = ((__builtin_amdgcn_s_memtime() - ct0) << 20) / (__builtin_amdgcn_s_memrealtime() - rt0 + 1);   // shader cycles per 100 MHz tick, x 2^20
    }
  }
}

}  // namespace w16

bool conv3x3_w16_eligible(const ConvArgs& a, int dtype) {
  return dtype == SS4K_F16 && a.w16 && a.epi == EPI_NHWC && !a.bsvd_resid && !a.dbg && !a.ups2 && a.cout_pad >= 64 && a.cout_pad % 64 == 0 &&
         (a.nchunks0 + a.nchunks1) % 2 == 0 && conv_plane_span_f16(a) < 4294967296.0;
}

void launch_conv3x3_w16(ss4k_ctx* ctx, const ConvArgs& a0, hipStream_t st) {
  using namespace w16;
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int groups = a.cout_pad / 64;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 / groups * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  // conv5 of an RDB with its residual through the matrix core (RL): res1 must be the conv's own input tensor = its first four planes
  // ... and 1 / alpha must be an fp16 number (the identity fragment carries it: RRDBNet's 0.2 -> 5.0); any other alpha would scale the skip
  // tensor by 1 +- 2^-11 on top of the output rounding, so it takes the epilogue that reads the residual from memory
  const bool alpha_exact = a.alpha != 0.f && std::fabs(__half2float(__float2half(1.f / a.alpha)) * a.alpha - 1.f) <= 1.2e-7f;
  const bool rl = a.wide_rl && a.res1 && a.act == ACT_NONE && alpha_exact && a.nchunks0 == 4 && a.cout_pad == 64 &&
                  a.res1 + (size_t)a.r1_plane0 * a.r1_plane_bytes == a.in0 + (size_t)a.in0_plane0 * a.in0_plane_bytes &&
                  a.r1_plane_bytes == a.in0_plane_bytes;
#ifdef SS4K_DEV
  static const bool stamp_mode = std::getenv("SS4K_W16_STAMP") && std::getenv("SS4K_W16_STAMP")[0] == '1';
  if (stamp_mode) {   // cycle counters of every wave of the first 1024 workgroups of cout group 0
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) SS4K_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 4 * 8 * 8));
    SS4K_HIP(hipMemsetAsync(dbuf, 0, 1024 * 4 * 8 * 8, st));
    a.dbg_buf = dbuf;
    const void* fs = rl ? reinterpret_cast<const void*>(&conv3x3_w16_kernel<true, true>) : reinterpret_cast<const void*>(&conv3x3_w16_kernel<true, false>);
    if (ctx->lds_attr_set.insert(fs).second) SS4K_HIP(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    if (rl) hipLaunchKernelGGL((conv3x3_w16_kernel<true, true>), dim3(gx, groups), dim3(64 * NW), LDS_BYTES, st, a);
    else hipLaunchKernelGGL((conv3x3_w16_kernel<true, false>), dim3(gx, groups), dim3(64 * NW), LDS_BYTES, st, a);
    SS4K_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hb(1024 * 4 * 8);
    SS4K_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
    static int printed = 0;
    if (printed++ < 4) {
      double acc[6] = {0}, tiles = 0, clk = 0; int nw = 0;
      for (size_t i = 0; i < 1024 * 4; ++i) {
        const unsigned long long* o = &hb[i * 8];
        if (!o[6]) continue;
        for (int k = 0; k < 6; ++k) acc[k] += (double)o[k];
        tiles += (double)o[6]; clk += (double)o[7] / 1048576.0 * 100.0; ++nw;
      }
      if (tiles > 0) {
        double tot = 0; for (double v : acc) tot += v;
        std::fprintf(stderr, "[w16 K=%d] %.0f MHz, cycles per tile and wave: total %.0f | init %.0f | phases 0+2 %.0f | phase 1 %.0f | vmcnt wait %.0f | barrier %.0f | epilogue %.0f\n",
                     a.nchunks0 + a.nchunks1, clk / nw, tot / tiles, acc[0] / tiles, acc[1] / tiles, acc[2] / tiles, acc[3] / tiles, acc[4] / tiles, acc[5] / tiles);
      }
    }
    return;
  }
#endif
  const void* fn = rl ? reinterpret_cast<const void*>(&conv3x3_w16_kernel<false, true>) : reinterpret_cast<const void*>(&conv3x3_w16_kernel<false, false>);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  ctx->prof_family = rl ? "w16::conv3x3_w16_kernel<RL> (64-cout tile, 16x16x32 MFMA, residual through the matrix core: conv5 of an RDB)"
                        : "w16::conv3x3_w16_kernel (64-cout tile, 16x16x32 MFMA)";
  if (rl) hipLaunchKernelGGL((conv3x3_w16_kernel<false, true>), dim3(gx, groups), dim3(64 * NW), LDS_BYTES, st, a);
  else hipLaunchKernelGGL((conv3x3_w16_kernel<false, false>), dim3(gx, groups), dim3(64 * NW), LDS_BYTES, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
