// 3x3 conv, fp16, 32 output channels, plain epilogue without residuals (conv1-4 of every RDB: 54 % of the headline's GPU time):
// conv_mfma.hip's <__half, 1, 4, 4> tile body with a THREE-stage ring of halo tiles.
//
// Why.  The LDS-weights kernel double-buffers a K-chunk: the halo tile of chunk c + 1 is in flight while chunk c computes, and the
// DMA instructions are spread over chunk c's MFMA stream - so the last of them are issued a few hundred cycles before the wait at
// the chunk's end and their round trip (L2 miss: Infinity Cache / HBM) is exposed.  The counters of the production build say the
// waves are parked 31 % of their time (SQ_WAIT_ANY: that wait and the barrier behind it) with the matrix pipe busy 2 x 27 %.
// Here the halo tile of chunk c + 2 is issued during chunk c (a whole chunk of slack for the round trip), the weights - L2-resident,
// short latency - stay one chunk ahead: three 19.1 KB tile stages + two 9 KB weight stages = 77 KB, still two workgroups per CU.
// The wait at a chunk's end is COUNTED: `vmcnt(tile instructions issued in this chunk)` lets exactly those stay in flight (loads,
// LDS-DMA and stores retire in issue order, so everything older - chunk c + 1's tile, its weights, the previous tile's stores - is
// covered).  One barrier per chunk as before: it hands stage (c - 1) % 3 and weight buffer (c - 1) % 2 to the DMA of the next chunk.
// Tile geometry, LDS image, swizzle, MFMA order and epilogue arithmetic are conv_mfma.hip's: results are bit-identical.
//
// MEASURED (round 3, tools/env_ab.py "SS4K_S3=0;SS4K_S3=1;SS4K_S3=0,SS4K_MB=4", headline job): 115.7 frames/s against 115.1 for the
// two-stage kernel on the same 16-row tiles and 117.1 on 20-row tiles (whose three stages would be 90 KB: one workgroup per CU).
// A whole chunk of extra slack for the round trip buys 0.5 %: the waves are not waiting for LATENCY.  Kept in the dev library
// (libss4k_hip_dev.so, SS4K_S3=1) as the experiment it is; the product library does not contain it.
#include "common.h"
#include "conv_tile.h"
#ifdef SS4K_DEV

namespace ss4k {
namespace s3 {

constexpr int NW = 4, MB = 4;
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

constexpr size_t lds_bytes_s3() { return (size_t)(3 * Geo<__half, MB, NW>::TILE_SLOTS + 2 * 9 * 64) * 16 + 64 * 4; }

__global__ __launch_bounds__(64 * NW, 2) void conv3x3_s3_kernel(const ConvArgs a) {
  using T = __half;
  using G = Geo<T, MB, NW>;
  constexpr int SPR = G::SPR, REC = G::REC, NG = 3;
  constexpr int TH = G::TH, TILE_SLOTS = G::TILE_SLOTS, TILE_DMA = G::TILE_DMA, DMA_PER_WAVE = G::DMA_PER_WAVE;
  constexpr int WSLOTS = 9 * 64;
  constexpr int TILE_BYTES = TILE_SLOTS * 16, W_BYTES = WSLOTS * 16;
  static_assert(TILE_DMA == NW * DMA_PER_WAVE, "every wave issues the same number of tile DMA instructions (counted waits)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [tile stage 0][1][2][weights 0][weights 1][32 bias + 32 slopes]
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  float* epi_lds = reinterpret_cast<float*>(smem + 3 * TILE_BYTES + 2 * W_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nchunks = a.nchunks0 + a.nchunks1;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = reinterpret_cast<const char*>(a.wpk);

  const bool banded = (gridDim.x % 8 == 0) && ntiles >= (int)gridDim.x;
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
  auto swz = [](int x) { return (x >> 3) & 1; };
  int rd_base[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = lr + dx;
    rd_base[dx] = (((wave * MB) * IN_W + x) * SPR + (lh ^ swz(x))) * 16;
  }
  int plan[DMA_PER_WAVE];
#pragma unroll
  for (int j = 0; j < DMA_PER_WAVE; ++j) {
    const int s = (wave + NW * j) * 64 + lane;
    const int p = s / SPR, gq = s % SPR;
    const int row = p / IN_W, x = p - row * IN_W;
    plan[j] = (s < TILE_SLOTS) ? (row | (x << 8) | (((gq ^ swz(x)) * 16) << 16)) : -1;
  }
  uint32_t src_off[DMA_PER_WAVE];
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
      const int iy = y0 - 1 + (plan[j] & 0xff), ix = x0 - 1 + ((plan[j] >> 8) & 0xff);
      const bool ok = plan[j] >= 0 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      src_off[j] = ok ? (uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix : OOB;
    }
  };
  auto plane_of = [&](int c) -> const char* {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  // per chunk and wave: NDMA_W weight instructions (for the NEXT chunk) first, then NDMA_T tile instructions (for the chunk after)
  constexpr int NDMA_T = DMA_PER_WAVE, NDMA_W = (9 + NW - 1) / NW, NDMA = NDMA_T + NDMA_W;
  constexpr int SE = NDMA <= NG * MB ? 3 : (2 * NDMA <= NG * 3 * MB ? 2 : 1);
  static_assert((3 * MB) % SE == 0 && NDMA <= NG * 3 * MB / SE, "not enough DMA slots in the MFMA stream");
  const char* pt_plane = nullptr; const char* pw_src = nullptr;
  uint32_t pt_dst = 0, pw_dst = 0; bool pt_on = false, pw_on = false;
  auto tile_begin = [&](int c, int stage) { pt_plane = plane_of(c); pt_dst = lds0 + stage * TILE_BYTES; pt_on = true; };
  auto weights_begin = [&](int c, int wbuf) { pw_src = wbase + (size_t)c * W_BYTES + lane * 16; pw_dst = lds0 + 3 * TILE_BYTES + wbuf * W_BYTES; pw_on = true; };
  auto dma_op = [&](int idx) {
    if (idx < NDMA_W) {
      const int k = wave + NW * idx;
      if (pw_on && k < 9) dma16(pw_src + k * 1024, __builtin_amdgcn_readfirstlane(pw_dst + k * 1024));
    } else if (idx < NDMA) {
      const int j = idx - NDMA_W, k = wave + NW * j;
      if (pt_on) {
        const size_t boff = (size_t)src_off[j] * REC + (size_t)((plan[j] >> 16) & 0xff);
        const char* src = src_off[j] != OOB ? pt_plane + boff : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(pt_dst + k * 1024);
        if (plan[j] >= 0) dma16(src, dst);   // (the last instruction of a tile has 8 live lanes: it is issued by every wave that owns one)
      }
    }
  };

  if (tid < 32) {
    epi_lds[tid] = tid < a.cout_pad ? a.bias[tid] : 0.f;
    epi_lds[32 + tid] = a.act == ACT_PRELU ? (tid < a.cout_pad ? a.prelu[tid] : 1.f) : (a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU6 ? 0.f : 1.f));
  }

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  // prologue: chunk 0 (weights + tile) and the tile of chunk 1
  weights_begin(0, 0); tile_begin(0, 0);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  pw_on = false; tile_begin(1, 1);
#pragma unroll
  for (int i = NDMA_W; i < NDMA; ++i) dma_op(i);
  wait_vm<NDMA_T>();   // chunk 0 has landed; chunk 1's tile may still be in flight
  __syncthreads();
  int ts = 0, wsel = 0;   // tile stage / weight buffer of the chunk being computed

  struct Frags { uint4 wf[3]; uint4 af[MB + 2]; };

  while (true) {
    f32x16 acc[MB];
    {
      float bias_v[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 b4 = *reinterpret_cast<const float4*>(epi_lds + 16 * (qd >> 1) + 8 * lh + 4 * (qd & 1));
        bias_v[4 * qd] = b4.x; bias_v[4 * qd + 1] = b4.y; bias_v[4 * qd + 2] = b4.z; bias_v[4 * qd + 3] = b4.w;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] = bias_v[i];
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    const int xo = cur_x0 + lr;

    for (int c = 0; c < nchunks; ++c) {
      // what goes in flight during this chunk: the weights of the next chunk, the halo tile of the one after
      pw_on = pt_on = false;
      if (c + 1 < nchunks) weights_begin(c + 1, wsel ^ 1);
      else if (next_tile >= 0) weights_begin(0, wsel ^ 1);
      const int st2 = ts >= 1 ? ts - 1 : 2;   // (ts + 2) % 3
      if (c + 2 < nchunks) tile_begin(c + 2, st2);
      else if (next_tile >= 0) {
        if (c + 2 == nchunks) setup_tile(next_tile, n, y0, x0);   // this tile's own halo tiles are all in flight or landed
        tile_begin(c + 2 - nchunks, st2);
      }
      const bool counted = pt_on;
      const char* tb = smem + ts * TILE_BYTES;
      const char* wb = smem + 3 * TILE_BYTES + wsel * W_BYTES + lane * 16;
      Frags f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) f.wf[dy] = *reinterpret_cast<const uint4*>(wb + (dy * 64) * 16);
#pragma unroll
      for (int ir = 0; ir < MB + 2; ++ir) f.af[ir] = *reinterpret_cast<const uint4*>(tb + rd_base[0] + ir * IN_W * REC);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bool more = g + 1 < NG;
        int m = 0;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int mb = ir - dy;
            if (mb >= 0 && mb < MB) {
              acc[mb] = mma<T>(f.wf[dy], f.af[ir], acc[mb]);
              if (m % SE == SE / 2 && g * (3 * MB / SE) + m / SE < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(g * (3 * MB / SE) + m / SE);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
          if (more) {
            __builtin_amdgcn_sched_barrier(0);
            f.af[ir] = *reinterpret_cast<const uint4*>(tb + rd_base[g + 1] + ir * IN_W * REC);
            if (ir >= MB - 1) {
              const int dy = ir - (MB - 1);
              f.wf[dy] = *reinterpret_cast<const uint4*>(wb + (((g + 1) * 3 + dy) * 64) * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      if (c + 1 < nchunks || next_tile >= 0) {
        // the next chunk's tile (issued a chunk ago) and weights (issued first in this chunk) have landed; this chunk's own
        // tile instructions - the youngest NDMA_T of this wave - stay in flight
        if (counted) wait_vm<NDMA_T>(); else wait_vm<0>();
        __syncthreads();
        ts = ts == 2 ? 0 : ts + 1; wsel ^= 1;
      }
    }

    // ---------------- epilogue: conv_mfma.hip's activation-only fast path (the LDS buffers were handed on above)
    {
      constexpr int HB = 16;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));
      const float alpha = a.alpha;
      float slope_v[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + 32 + 16 * (qd >> 1) + 8 * lhe + 4 * (qd & 1));
        slope_v[4 * qd] = s4.x; slope_v[4 * qd + 1] = s4.y; slope_v[4 * qd + 2] = s4.z; slope_v[4 * qd + 3] = s4.w;
      }
      const size_t sub = (size_t)lhe * HB;
      char* outp = a.out + (size_t)a.out_plane0 * a.out_plane_bytes + sub;
      const size_t pix0 = ((size_t)cur_n * a.H + cur_y0 + wave * MB) * a.W + xo;
      const bool select_form = a.act == ACT_PRELU;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const bool ok = (cur_y0 + wave * MB + mb) < a.H && xo < a.W;
        float v[16];
        if (a.act == ACT_RELU6) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fminf(fmaxf(acc[mb][i], 0.f), 6.f) * alpha;
        } else if (select_form) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float t = acc[mb][i], neg = t * slope_v[i];
            v[i] = (t >= 0.f ? t : neg) * alpha;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float t = acc[mb][i];
            v[i] = fmaxf(t, t * slope_v[i]) * alpha;
          }
        }
        if (ok) {
          char* o = outp + (pix0 + (size_t)mb * a.W) * REC;
          store8<T>(o, v);
          store8<T>(o + a.out_plane_bytes, v + 8);
        }
      }
    }
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
}

}  // namespace s3

bool conv3x3_s3_eligible(const ConvArgs& a, int dtype) {
  return dtype == SS4K_F16 && a.cout_pad == 32 && a.epi == EPI_NHWC && !a.bsvd_resid && !a.res1 && !a.res2 && !a.ups2 && !a.dbg &&
         a.nchunks0 + a.nchunks1 >= 2;
}

void launch_conv3x3_s3(ss4k_ctx* ctx, const ConvArgs& a0, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + 16 - 1) / 16;
  a.zero_page = ctx->zero_page();
  constexpr size_t lds = s3::lds_bytes_s3();
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  const void* fn = reinterpret_cast<const void*>(&s3::conv3x3_s3_kernel);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  hipLaunchKernelGGL(s3::conv3x3_s3_kernel, dim3(gx), dim3(64 * s3::NW), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace ss4k
#endif  // SS4K_DEV
