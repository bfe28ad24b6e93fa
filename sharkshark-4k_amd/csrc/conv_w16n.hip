// conv3x3_w16n_kernel: a 3x3 layer with at most 16 output channels and the final NCHW fp32 hand-off (RRDBNet's conv_last: 64 -> 3 at the
// output resolution, 2.3 % of a 720p -> 1440p job) on conv_w16.hip's machinery with ONE 16-cout MFMA block.
//
// On conv_mfma.hip's 32-cout tile the layer computes 32 output channels for 3; v_mfma_f32_16x16x32_f16 has M = 16: half the matrix work,
// and what is left is the layer's real bound - 128 bytes of input records per output pixel.  Same phases as conv_w16.hip (a pair of
// K-chunks = plane 2q taps dx 0 | 1, both planes dx 2, plane 2q + 1 taps dx 0 | 1), same 18 x 34 halo tiles in two fixed buffers, same
// request schedule; a phase's weights are [dy][lane] = 3 KB (ring of three); a wave owns 4 rows x 32 pixels: 32 accumulator registers,
// 48.6 KB of LDS and <= 168 registers: THREE workgroups per CU.  The bias is the C operand of an accumulator's first MFMA.  Epilogue: the
// lanes of row group 0 hold output channels 0..3 of their pixel: one 4-byte store per real channel into its fp32 plane.
#include "common.h"
#include "conv_tile.h"
#include <type_traits>

namespace ss4k {
namespace w16n {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NW = 4, MB = 4, TH = NW * MB;
constexpr int XH = TH + 2, XW = TW + 2;
constexpr int REC = 32;
constexpr int ROWX = XW * REC;
constexpr int XT_SLOTS = XH * XW * 2;               // 1224
constexpr int XT_BYTES = XT_SLOTS * 16;             // 19584
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 20
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 5
constexpr int WP = 3 * 1024, NWS = 3;               // a phase's weights [dy][lane], ring slots
constexpr int NDMA = DMA_PER_WAVE + 1;              // per wave and phase: 5 halo-tile pieces, one of the 3 weight pieces (waves 0..2)
constexpr int W_OFF = 2 * XT_BYTES, B_OFF = W_OFF + NWS * WP;
constexpr size_t LDS_BYTES = B_OFF + 16 * 4;
static_assert(3 * LDS_BYTES <= 160 * 1024, "three workgroups per CU");

__device__ __forceinline__ f32x4 mma16(const uint4& w, const uint4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}

__global__ __launch_bounds__(64 * NW, 3) void conv3x3_w16n_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row group: tile rows 4 wave .. 4 wave + 3
  const int n16 = lane & 15, kg = lane >> 4;
  const int K = a.nchunks0 + a.nchunks1, NP = K >> 1;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = reinterpret_cast<const char*>(a.w16);

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
  int rdA[2], rdX[2];
#pragma unroll
  for (int hn = 0; hn < 2; ++hn) {
    rdA[hn] = (((wave * MB) * XW + 16 * hn + n16 + (kg >> 1)) * 2 + (kg & 1)) * 16;
    rdX[hn] = (((wave * MB) * XW + 16 * hn + n16 + 2) * 2 + (kg & 1)) * 16 + (kg >> 1) * XT_BYTES;
  }
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };
  auto tile_src = [&](int k, int n, int y0, int x0) __attribute__((always_inline)) -> uint32_t {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // computed where it is used (conv_d16.hip)
    const int s = k * 64 + ln;
    const int p = s >> 1, gq = s & 1;
    const int row = p / XW, x = p - row * XW;
    const int iy = y0 - 1 + row, ix = x0 - 1 + x;
    const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    return ok ? ((uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix) * REC + (uint32_t)(gq * 16) : OOB;
  };
  auto plane_of = [&](int c) __attribute__((always_inline)) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  const char* pf_plane = nullptr; const char* pf_w = nullptr; int pf_buf = 0, pf_slot = 0, pf_n = 0, pf_y0 = 0, pf_x0 = 0; bool pf_tile = false, pf_wt = false;
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
      int l16 = lane * 16;
      asm volatile("" : "+v"(l16));
      if (pf_wt && wave < 3) dma16(pf_w + wave * 1024 + l16, __builtin_amdgcn_readfirstlane(lds0 + W_OFF + pf_slot * WP + wave * 1024));
    }
  };
  auto slot_inc = [](int s) { return s == NWS - 1 ? 0 : s + 1; };

  float* bias_lds = reinterpret_cast<float*>(smem + B_OFF);
  if (tid < 16) bias_lds[tid] = tid < a.cout_pad ? a.bias[tid] : 0.f;

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  pf_n = n; pf_y0 = y0; pf_x0 = x0;
  int ws = 0;
  pf_tile = true; pf_plane = plane_of(0); pf_buf = 0; pf_wt = true; pf_w = wbase; pf_slot = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  pf_tile = false; pf_w = wbase + WP; pf_slot = 1;
  dma_op(DMA_PER_WAVE);
  dma_wait();
  __syncthreads();
  const int lane16 = lane * 16;
  uint4 wf[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) wf[dy] = *reinterpret_cast<const uint4*>(smem + W_OFF + lane16 + dy * 1024);

  while (true) {
    f32x4 acc[MB][2];
    f32x4 bias4;
    {
      const float4* bp = reinterpret_cast<const float4*>(smem + B_OFF + kg * 16);
      int o = 0;
      asm volatile("" : "+v"(o));
      const float4 b0 = bp[o];
      bias4 = f32x4{b0.x, b0.y, b0.z, b0.w};
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);

    auto period = [&](const int q, auto FIRSTP) __attribute__((always_inline)) {
      const bool last_q = q + 1 == NP;
      auto phase = [&](auto PH, auto FIRST) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value;
        constexpr bool first = decltype(FIRST)::value;
        const int lp2 = 3 * q + ph + 2;
        if (lp2 < 3 * NP) { pf_wt = true; pf_w = wbase + (size_t)lp2 * WP; }
        else if (next_tile >= 0) { pf_wt = true; pf_w = wbase + (size_t)(lp2 - 3 * NP) * WP; }
        else pf_wt = false;
        pf_slot = ws == 0 ? NWS - 1 : ws - 1;
        if constexpr (ph == 0) { pf_tile = true; pf_plane = plane_of(2 * q + 1); pf_buf = 1; }
        else if constexpr (ph == 1) { pf_tile = false; }
        else {
          pf_buf = 0;
          if (!last_q) { pf_tile = true; pf_plane = plane_of(2 * q + 2); }
          else if (next_tile >= 0) { setup_tile(next_tile, n, y0, x0); pf_n = n; pf_y0 = y0; pf_x0 = x0; pf_tile = true; pf_plane = plane_of(0); }
          else pf_tile = false;
        }
        const bool more = !(last_q && ph == 2) || next_tile >= 0;
        const char* tb = smem + (ph == 2 ? XT_BYTES : 0);
        const int* rd = ph == 1 ? rdX : rdA;
        const char* wbn = smem + W_OFF + slot_inc(ws) * WP + lane16;
        uint4 bf[3][2];
        auto bf_load = [&](int t, int hn) __attribute__((always_inline)) { return *reinterpret_cast<const uint4*>(tb + rd[hn] + t * ROWX); };
#pragma unroll
        for (int t = 0; t < 3; ++t) { bf[t][0] = bf_load(t, 0); bf[t][1] = bf_load(t, 1); }
        int m = 0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int r = ir - dy;
            if (r >= 0 && r < MB) {
#pragma unroll
              for (int hn = 0; hn < 2; ++hn) {
                acc[r][hn] = mma16(wf[dy], bf[ir % 3][hn], (first && dy == 0) ? bias4 : acc[r][hn]);
                // 24 MFMAs per phase: a DMA slot after every fourth (five halo-tile pieces, one weight piece)
                if ((m & 3) == 3) {
                  __builtin_amdgcn_sched_barrier(0);
                  dma_op(m >> 2);
                  __builtin_amdgcn_sched_barrier(0);
                }
                ++m;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (ir + 3 < MB + 2) { bf[ir % 3][0] = bf_load(ir + 3, 0); bf[ir % 3][1] = bf_load(ir + 3, 1); }
          if (more && ir >= MB - 1) wf[ir - (MB - 1)] = *reinterpret_cast<const uint4*>(wbn + (ir - (MB - 1)) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) { dma_wait(); __syncthreads(); }
        ws = slot_inc(ws);
      };
      phase(std::integral_constant<int, 0>{}, FIRSTP);
      phase(std::integral_constant<int, 1>{}, std::false_type{});
      phase(std::integral_constant<int, 2>{}, std::false_type{});
    };
    period(0, std::true_type{});
#pragma unroll 1
    for (int q = 1; q < NP; ++q) period(q, std::false_type{});

    // ---------------- epilogue: NCHW fp32; row group 0's lanes hold output channels 0..3 of their pixel
    if (kg == 0) {
      float* o = reinterpret_cast<float*>(a.out);
      const size_t plane = (size_t)a.H * a.W;
#pragma unroll
      for (int r = 0; r < MB; ++r) {
        const int y = cur_y0 + wave * MB + r;
#pragma unroll
        for (int hn = 0; hn < 2; ++hn) {
          const int x = cur_x0 + 16 * hn + n16;
          if (y < a.H && x < a.W) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (i < a.cout_real) o[((size_t)cur_n * a.cout_real + i) * plane + (size_t)y * a.W + x] = acc[r][hn][i];
          }
        }
      }
    }
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
}

}  // namespace w16n

// a layer for the narrow build: fp16, <= 4 real output channels handed over as NCHW fp32, no activation / residual, an even number of K-chunks
bool conv3x3_w16n_eligible(const ConvArgs& a, int dtype) {
  return dtype == SS4K_F16 && a.w16 && a.epi == EPI_NCHW_F32 && a.cout_real <= 4 && a.act == ACT_NONE && !a.res1 && !a.res2 && !a.bsvd_resid &&
         a.alpha == 1.f && !a.dbg && !a.ups2 && (a.nchunks0 + a.nchunks1) % 2 == 0 && conv_plane_span_f16(a) < 4294967296.0;
}

void launch_conv3x3_w16n(ss4k_ctx* ctx, const ConvArgs& a0, hipStream_t st) {
  using namespace w16n;
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 3 * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  const void* fn = reinterpret_cast<const void*>(&conv3x3_w16n_kernel);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
  ctx->prof_family = "w16n::conv3x3_w16n_kernel (one 16-cout block, 16x16x32 MFMA, NCHW fp32 hand-off: a final 64 -> 3 layer)";
  hipLaunchKernelGGL(conv3x3_w16n_kernel, dim3(gx), dim3(64 * NW), LDS_BYTES, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
