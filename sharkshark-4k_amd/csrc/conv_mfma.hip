// 3x3 convolution (pad 1, stride 1) as an im2col-free implicit GEMM on the CDNA4 matrix cores.
//
//   D[cout][pixel] += W[cout][k] * X[k][pixel],   k = (tap, cin)
//
// Data layout in HBM ("planes"): an activation tensor with C channels is stored as C/16 planes of
// N*H*W records of 16 channels (32 bytes fp16, 64 bytes fp32), so one K-chunk (= one plane) of one
// image row is a contiguous run of bytes, the RRDB dense concat is just "more planes", and every
// load/store is a whole record.  K-chunks of 16 channels keep a pipeline stage small (halo tile +
// weights of a 16x32x64 tile: 38 KB), which is what lets the pixel tile per weight fetch be large.
//
// A workgroup is NW waves, persistent over an XCD-banded walk of output tiles (production fp16
// builds: 4 waves, TWO workgroups per CU so one's epilogue / barrier / DMA wait hides under the
// other's MFMAs):
//   * output tile (NW*MB) rows x 32 pixels x (NB*32) output channels; wave w owns rows MB*w...;
//   * per K-chunk the halo tile ((NW*MB+2) x 34 records) and the chunk's weights (9*NB KB fp16,
//     pre-packed on the host in MFMA fragment order) are DMA'd global->LDS with
//     global_load_lds_dwordx4 into the buffer the waves are NOT computing from (two LDS buffers,
//     one barrier per chunk; the DMA instructions of chunk c+1 - or of the next tile's first chunk
//     - are issued from slots inside the MFMA stream of chunk c);
//   * the tile image in LDS is pixel-major with a pixel's 16-byte slots XOR-swizzled by its x,
//     applied on the DMA *source* address, so every ds_read_b128 operand read is bank-conflict free;
//   * loop nest per chunk: per (dx, k-step) group 3 weight fragments (dy = 0..2) stay in registers
//     while the wave walks its MB+2 input rows - each activation fragment is read ONCE and feeds up
//     to three MFMAs; the MB = 4 builds keep one fragment set and refill registers in place;
//   * A operand = weights, B operand = activations; the MFMA row -> cout map (host-side weight
//     permutation) gives lane (pixel, half h) channels 8h..8h+7 of the block's two 16-channel
//     planes: bias / activation / residual epilogue in registers, and every store instruction of a
//     wave writes one contiguous kilobyte (non-temporal);
//   * the LDS buffers are handed to the next tile BEFORE the epilogue, so output stores drain
//     under the next chunk's MFMAs instead of in front of a vmcnt(0);
//   * f16: v_mfma_f32_32x32x16_f16 (fp32 accumulate); f32: v_mfma_f32_32x32x2_f32 (exact fp32).
// Nearest-x2 upsampled input, stride-2 subsampling, PixelShuffle(2) and the NCHW fp32 hand-off are
// address modes of the DMA source / the epilogue store, not extra passes.
#include "common.h"
#include "conv_tile.h"

#ifndef SS4K_ROLL
#define SS4K_ROLL(MB) ((MB) >= 4)
#endif

namespace ss4k {

int conv_cw(int) { return CW; }

// EK = epilogue kind the build is specialised for (a runtime switch between them costs registers - the all-in-one
// build of round 1 spilled 52-68 bytes - and the kinds differ in the MFMA loop as well):
//   EK_PLAIN   plain layout (every RRDBNet / SRVGG body layer, most BSVD layers): branch-free fast path
//   EK_SUB2    stride-2 conv (BSVD downc0 / downc1): only the even output rows are computed (half the MFMAs of the
//              stride-1 form), even pixels stored
//   EK_PS2     PixelShuffle(2) (BSVD upc2 / upc1): the virtual cout order (pack.cpp) puts the two horizontal
//              sub-pixels of an output row into the two planes of a 32-cout block, so a wave's two store
//              instructions write one contiguous run of output records
//   EK_GENERAL fp32 NCHW hand-off (network outputs), BSVD residual, anything else
enum { EK_PLAIN = 0, EK_SUB2 = 1, EK_PS2 = 2, EK_GENERAL = 3 };
template <typename T, int NB, int MB, int NW, int DBG, int EK>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 2 : wgs_per_cu<T, NB, MB, NW>())) void conv3x3_kernel(const ConvArgs a) {
  using G = Geo<T, MB, NW>;
  constexpr int KS = G::KS, SPR = G::SPR, REC = G::REC, NG = 3 * KS;
  constexpr int TH = G::TH, TILE_SLOTS = G::TILE_SLOTS, TILE_DMA = G::TILE_DMA;
  constexpr int DMA_PER_WAVE = G::DMA_PER_WAVE;
  constexpr int WSLOTS = 9 * KS * NB * 64;  // weight slots per chunk
  constexpr bool ROLL = SS4K_ROLL(MB);       // one fragment set reloaded in place (see the chunk loop)
  constexpr int RV = (int)(16 * sizeof(T) / 16);  // uint4 per 16-channel group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [tile buf 0][tile buf 1][weights buf 0][weights buf 1]
  constexpr int TILE_BYTES = TILE_SLOTS * 16, W_BYTES = WSLOTS * 16;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = blockIdx.y;
  const int nchunks = a.nchunks0 + a.nchunks1;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = reinterpret_cast<const char*>(a.wpk) + (size_t)grp * nchunks * W_BYTES;
  const int Hs = a.ups2 ? (a.H >> 1) : a.H, Ws = a.ups2 ? (a.W >> 1) : a.W;

  // XCD-aware persistent tile walk: workgroups b and b+8 share an XCD (and its L2); give each XCD
  // a contiguous band of tiles so halo rows and the planes a layer just wrote are re-read from
  // the same L2.  Placement only changes speed, never results.
  const bool banded = (gridDim.x % 8 == 0) && ntiles >= (int)gridDim.x && !a.no_band;
  const int tpx = (ntiles + 7) / 8;
  // a.reverse walks the same tiles back to front (inside each XCD's band, so the band -> XCD map
  // stays): consecutive layers alternate direction, and a layer starts on the planes the previous
  // one touched last - still in that XCD's L2 / the Infinity Cache
  auto tile_of = [&](int k) -> int {
    if (!banded) {
      const int t = blockIdx.x + k * gridDim.x;
      return t < ntiles ? (a.reverse ? ntiles - 1 - t : t) : -1;
    }
    const int base = (blockIdx.x & 7) * tpx, len = min(tpx, ntiles - base);
    const int j = (blockIdx.x >> 3) + k * (gridDim.x >> 3);
    return j < len ? base + (a.reverse ? len - 1 - j : j) : -1;
  };

  // bank swizzle of a pixel's 16-byte slots by its x: makes every ds_read_b128 operand read
  // conflict-free for 64-byte (4 slots) and 32-byte (2 slots) pixels alike
  auto swz = [](int x) { return SPR == 4 ? ((x >> 2) & 3) : ((x >> 3) & 1); };
  // per-lane operand read base (bytes): slot = pixel*SPR + ((2ks+lh) ^ swz(x)), x = lr+dx
  int rd_base[3][KS];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int x = lr + dx;
      rd_base[dx][ks] = (((wave * MB) * IN_W + x) * SPR + ((2 * ks + lh) ^ swz(x))) * 16;
    }

  // tile-independent part of the DMA plan: LDS slot s = 64k + lane holds pixel (row, x), source
  // channel group gq ^ swz(x)
  int plan[DMA_PER_WAVE];  // row | x << 8 | group*16 << 16 | valid << 31
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
      const int sy = a.ups2 ? (iy >> 1) : iy, sx = a.ups2 ? (ix >> 1) : ix;
      src_off[j] = ok ? (uint32_t)(n * Hs + sy) * (uint32_t)Ws + (uint32_t)sx : OOB;  // source pixel index inside a plane
    }
  };
  // One K-chunk prefetch = NDMA wave-level DMA instructions per wave (halo tile, then weights).
  // They are not issued in a burst: dma_op(i) is called from slots spread through the MFMA stream
  // of the chunk being computed, so their issue cost hides under matrix-pipe time.
  constexpr int NDMA_T = DMA_PER_WAVE, NDMA_W = (9 * KS * NB + NW - 1) / NW, NDMA = NDMA_T + NDMA_W;
  // one DMA slot after every SE-th (row, dy) MFMA pair of a group
  constexpr int SE = NDMA <= NG * MB ? 3 : (2 * NDMA <= NG * 3 * MB ? 2 : 1);
  static_assert((3 * MB) % SE == 0 && NDMA <= NG * 3 * MB / SE, "not enough DMA slots in the MFMA stream");
  const char* pf_plane = nullptr; const char* pf_wsrc = nullptr;
  uint32_t pf_tdst = 0, pf_wdst = 0; bool pf_on = false;
  auto prefetch_begin = [&](int c, int buf) {
    pf_plane = (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                                : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
    pf_tdst = lds0 + buf * TILE_BYTES;
    pf_wsrc = wbase + (size_t)c * W_BYTES + lane * 16;
    pf_wdst = lds0 + 2 * TILE_BYTES + buf * W_BYTES;
    pf_on = true;
  };
  auto dma_op = [&](int idx) {
    if (!pf_on) return;
    if (idx < NDMA_T) {
      const int k = wave + NW * idx;
      if (k < TILE_DMA && !(DBG & DBG_NO_TILE_DMA)) {
        const size_t boff = (size_t)src_off[idx] * REC + (size_t)((plan[idx] >> 16) & 0xff);  // record + swizzled 16-byte slot
        const char* src = (src_off[idx] != OOB && !(DBG & DBG_NO_MMA)) ? ((((DBG & DBG_NO_STORE) != 0) != ((DBG & DBG_NO_EPILOGUE) != 0)) ? a.in0 + (boff & 0x1FFFFFu) : pf_plane + boff) : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(pf_tdst + k * 1024);
        if (plan[idx] >= 0) dma16(src, dst);  // lanes past the tile's last slot are masked off (EXEC)
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - NDMA_T);
      if (k < 9 * KS * NB && !(DBG & DBG_NO_W_DMA)) dma16(((DBG & DBG_NO_MMA) ? wbase + lane * 16 : pf_wsrc + k * 1024), __builtin_amdgcn_readfirstlane(pf_wdst + k * 1024));
    }
  };

  // epilogue constants that do not depend on the tile (bias, negative slope per channel) live in LDS
  float* epi_lds = reinterpret_cast<float*>(smem + 2 * TILE_BYTES + 2 * W_BYTES);  // [NB*32][2]
  if (tid < NB * 32) {
    const int v = grp * NB * 32 + tid;
    epi_lds[tid] = v < a.cout_pad ? a.bias[v] : 0.f;
    epi_lds[NB * 32 + tid] = a.act == ACT_PRELU ? (v < a.cout_pad ? a.prelu[v] : 1.f) : (a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU6 ? 0.f : 1.f));
  }

  // DBG_STAMP build only: per-phase cycle totals of wave 0 (s_memtime), written once at the end
  unsigned long long st_dma = 0, st_mma = 0, st_epi = 0, st_bar = 0, st_t = 0, st_store = 0, st_wait = 0, st_rt0 = 0;
  if constexpr ((DBG & DBG_STAMP) != 0) st_rt0 = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&]() -> unsigned long long {
    if constexpr ((DBG & DBG_STAMP) != 0) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      return t;
    } else {
      return 0ull;
    }
  };
  const unsigned long long st_begin = stamp();

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  prefetch_begin(0, 0);
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  dma_wait();
  __syncthreads();
  int buf = 0;
  const unsigned long long st_pro = stamp() - st_begin;

  struct Frags { uint4 wf[3][NB]; uint4 af[MB + 2]; };

  while (true) {
    f32x16 acc[NB][MB];  // start from the bias: one less pass over the tile in the epilogue
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float bias_v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(epi_lds + nb * 32 + 16 * (q >> 1) + 8 * lh + 4 * (q & 1));
        bias_v[4 * q] = b4.x; bias_v[4 * q + 1] = b4.y; bias_v[4 * q + 2] = b4.z; bias_v[4 * q + 3] = b4.w;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][mb][i] = bias_v[i];
    }

    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    const int xo = cur_x0 + lr;

    for (int c = 0; c < nchunks; ++c) {
      st_t = stamp();
      // put the next K-chunk (or the next tile's first chunk) in flight into the other buffer
      pf_on = false;
      if (c + 1 < nchunks) {
        prefetch_begin(c + 1, buf ^ 1);
      } else if (next_tile >= 0) {
        setup_tile(next_tile, n, y0, x0);
        prefetch_begin(0, buf ^ 1);
      }
      { const unsigned long long t = stamp(); st_dma += t - st_t; st_t = t; }
      const char* tb = smem + buf * TILE_BYTES;
      const char* wb = smem + 2 * TILE_BYTES + buf * W_BYTES + lane * 16;
      auto load_group = [&](Frags& f, int g) {
        const int dx = g / KS, ks = g % KS;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            f.wf[dy][nb] = *reinterpret_cast<const uint4*>(wb + (((g * 3 + dy) * NB + nb) * 64) * 16);
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir)
          f.af[ir] = *reinterpret_cast<const uint4*>(tb + rd_base[dx][ks] + ir * IN_W * REC);
      };
      // the 12 (input row, dy) pairs of one group, in an order that never repeats an accumulator
      // back to back; after every third MFMA one DMA instruction of the prefetch is issued
      auto mma_group = [&](const Frags& f, int g) {
        // the 3*MB (input row, dy) pairs of one group; after every third MFMA (x NB) one DMA
        // instruction of the prefetch is issued
        int m = 0;
        __builtin_amdgcn_s_setprio(1);  // keep this wave's MFMA burst together while its SIMD partner loads
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir)
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int mb = ir - dy;
            if (mb >= 0 && mb < MB) {
              if (!(EK == EK_SUB2 && (mb & 1))) {   // stride 2: odd output rows are never stored
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb][mb] = mma<T>(f.wf[dy][nb], f.af[ir], acc[nb][mb]);
              }
              if (m % SE == SE / 2 && g * (3 * MB / SE) + m / SE < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(g * (3 * MB / SE) + m / SE);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
        __builtin_amdgcn_s_setprio(0);
      };
      if constexpr (!ROLL) {
        // software pipeline over the NG (dx, k-step) groups: the LDS reads of group g+1 are in
        // flight while the 3*MB*NB MFMAs of group g issue (two fragment sets)
        Frags fr[2];
        load_group(fr[0], 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) load_group(fr[(g + 1) & 1], g + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma_group(fr[g & 1], g);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // ONE fragment set, reloaded in place: as soon as the last MFMA that reads a register of
        // group g has issued, that register is refilled with its group g+1 value (input row ir after
        // its <= 3 MFMAs; the dy weights after rows MB-1, MB, MB+1).  Same latency cover as two
        // sets for all but the last few MFMAs of a group, at half the registers - what lets the
        // 4-rows-per-wave builds keep two waves per SIMD without scratch.
        Frags f;
        load_group(f, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const bool more = g + 1 < NG;
          const int dxn = (g + 1) / KS, ksn = (g + 1) % KS;
          int m = 0;
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ir = 0; ir < MB + 2; ++ir) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
              const int mb = ir - dy;
              if (mb >= 0 && mb < MB) {
                if (!(EK == EK_SUB2 && (mb & 1))) {   // stride 2: odd output rows are never stored
#pragma unroll
                  for (int nb = 0; nb < NB; ++nb) acc[nb][mb] = mma<T>(f.wf[dy][nb], f.af[ir], acc[nb][mb]);
                }
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
              f.af[ir] = *reinterpret_cast<const uint4*>(tb + rd_base[dxn][ksn] + ir * IN_W * REC);
              if (ir >= MB - 1) {
                const int dy = ir - (MB - 1);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                  f.wf[dy][nb] = *reinterpret_cast<const uint4*>(wb + ((((g + 1) * 3 + dy) * NB + nb) * 64) * 16);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
        }
      }
      { const unsigned long long t = stamp(); st_mma += t - st_t; st_t = t; }
      if (c + 1 < nchunks) {
        dma_wait();       // next chunk has landed in the other buffer
        { const unsigned long long t = stamp(); st_wait += t - st_t; st_t = t; }
        __syncthreads();  // and every wave is done reading this one
        buf ^= 1;
        { const unsigned long long t = stamp(); st_bar += t - st_t; st_t = t; }
      }
    }

    // Hand the LDS buffers over to the next tile BEFORE the epilogue: vmcnt counts loads, stores and
    // LDS-DMA together in issue order, so a wait placed after the epilogue would also sit out the
    // round trip of this tile's output stores (measured: 18 % of the kernel).  Here only the DMA of
    // the next tile's first chunk is outstanding; the stores then drain under that chunk's MFMAs,
    // and the waves run their epilogues unsynchronised.  The epilogue touches no tile buffer.
    if (next_tile >= 0) {
      dma_wait();       // next tile's first chunk has landed
      { const unsigned long long t = stamp(); st_wait += t - st_t; st_t = t; }
      __syncthreads();  // all waves are done with the last buffer
      buf ^= 1;
      { const unsigned long long t = stamp(); st_bar += t - st_t; st_t = t; }
    }

    // ---------------- epilogue: bias, activation, residuals, layout-aware store ----------------
    // The RRDB output is written in place over res2, so the compiler cannot hoist residual loads
    // past the stores: per 32-channel block issue every row's residual loads first, then consume
    // (one memory round trip per block instead of one per row).
    // Lane (pixel p, half h) owns, per 32-cout block, channels 8h..8h+7 of its first plane
    // (accumulator elements 0-7) and of its second plane (elements 8-15): one store instruction of
    // the wave = plane X, 32 pixels x two 16-byte halves = one fully contiguous kilobyte.
    constexpr int HB = 8 * (int)sizeof(T);  // bytes of an 8-channel half record
    constexpr int RV8 = HB / 16;
    const bool batch_res = a.epi == EPI_NHWC && !a.bsvd_resid && (a.res1 || a.res2);
    // Fast path (every RRDBNet / SRVGG body layer): plain layout, branch-free arithmetic
    //   v = act(acc) * alpha + res1;  v = v * gamma + res2      (absent residuals are zeros,
    // absent activation is slope 1, so the same instruction stream serves every such layer)
    constexpr bool GEN = EK == EK_GENERAL;
    const bool fast_epi = EK == EK_PLAIN;
    // the epilogue's per-lane plane pointers depend only on kernel arguments: left alone, the
    // compiler computes them once before the tile loop and carries ~16 registers through the MFMA
    // loop (spilling in the 4-rows-per-wave builds).  Re-derive them per tile instead.
    int lhe = lh;
    asm volatile("" : "+v"(lhe));
    if (fast_epi) {
      const float alpha = a.alpha, gamma = a.gamma;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int vblock = (grp * NB + nb) * 32;
        if (vblock >= a.cout_pad) continue;
        float slope_v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + NB * 32 + nb * 32 + 16 * (q >> 1) + 8 * lhe + 4 * (q & 1));
          slope_v[4 * q] = s4.x; slope_v[4 * q + 1] = s4.y; slope_v[4 * q + 2] = s4.z; slope_v[4 * q + 3] = s4.w;
        }
        const int opl = vblock / CW;  // first of this block's two planes
        const size_t sub = (size_t)lhe * HB;
        const char* r1p = a.res1 ? a.res1 + (size_t)(a.r1_plane0 + opl) * a.r1_plane_bytes + sub : nullptr;
        const char* r2p = a.res2 ? a.res2 + (size_t)(a.r2_plane0 + opl) * a.r2_plane_bytes + sub : nullptr;
        char* outp = a.out + (size_t)(a.out_plane0 + opl) * a.out_plane_bytes + sub;
        const size_t pix0 = ((size_t)cur_n * a.H + cur_y0 + wave * MB) * a.W + xo;
        if (!r1p && !r2p) {
          // activation only (four of the five RDB convs): max-form LeakyReLU / identity when every
          // slope is in [0,1] (a.act != PRELU), select form for learned PReLU slopes
          const bool select_form = a.act == ACT_PRELU;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const bool ok = (cur_y0 + wave * MB + mb) < a.H && xo < a.W;
            float v[16];
            if (a.act == ACT_RELU6) {
#pragma unroll
              for (int i = 0; i < 16; ++i) v[i] = fminf(fmaxf(acc[nb][mb][i], 0.f), 6.f) * alpha;
            } else if (select_form) {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const float t = acc[nb][mb][i], neg = t * slope_v[i];
                v[i] = (t >= 0.f ? t : neg) * alpha;
              }
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                // max(t, slope t) in ONE instruction, v_med3_f32(t, slope t, +inf) (fmaxf is two in IEEE mode: the compiler first quiets
                // a possible signalling NaN with v_max(t, t)); the same value for every non-NaN t
                const float t = acc[nb][mb][i];
                v[i] = __builtin_amdgcn_fmed3f(t, t * slope_v[i], __builtin_inff()) * alpha;
              }
            }
            if (ok) {
              if constexpr ((DBG & DBG_NO_STORE) != 0) { if (v[0] == 12345.678f) a.out[0] = 1; }
              else {
                char* o = outp + (pix0 + (size_t)mb * a.W) * REC;
                store8<T>(o, v);
                store8<T>(o + a.out_plane_bytes, v + 8);
              }
            }
          }
        } else {
          // RB rows at a time: residual loads of those rows first, then the arithmetic (one memory
          // round trip per batch, and the 4-rows-per-wave builds stay inside their register budget)
          constexpr int RB = (NB * MB >= 8) ? 1 : 2;  // 128 accumulator registers leave room for one row only
#pragma unroll
          for (int mb0 = 0; mb0 < MB; mb0 += RB) {
            uint4 r1v[RB][2][RV8], r2v[RB][2][RV8];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
              const int mb = mb0 + j;
              const bool ok = (cur_y0 + wave * MB + mb) < a.H && xo < a.W;
              const size_t rec = (pix0 + (size_t)mb * a.W) * REC;
#pragma unroll
              for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                for (int q = 0; q < RV8; ++q) {
                  r1v[j][hq][q] = (r1p && ok) ? *reinterpret_cast<const uint4*>(r1p + hq * (size_t)a.r1_plane_bytes + rec + 16 * q) : make_uint4(0, 0, 0, 0);
                  r2v[j][hq][q] = (r2p && ok) ? *reinterpret_cast<const uint4*>(r2p + hq * (size_t)a.r2_plane_bytes + rec + 16 * q) : make_uint4(0, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < RB; ++j) {
              const int mb = mb0 + j;
              const bool ok = (cur_y0 + wave * MB + mb) < a.H && xo < a.W;
              float v[16], r1[16], r2[16];
#pragma unroll
              for (int hq = 0; hq < 2; ++hq) {
                load8<T>(reinterpret_cast<const char*>(&r1v[j][hq][0]), r1 + 8 * hq);
                load8<T>(reinterpret_cast<const char*>(&r2v[j][hq][0]), r2 + 8 * hq);
              }
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                float t = acc[nb][mb][i];
                const float neg = t * slope_v[i];
                t = t >= 0.f ? t : neg;
                if (a.act == ACT_RELU6) t = fminf(t, 6.f);  // slope is 0 for ReLU6: the select above is the ReLU
                t = t * alpha + r1[i];
                v[i] = t * gamma + r2[i];
              }
              if (ok) {
                if constexpr ((DBG & DBG_NO_STORE) != 0) { if (v[0] == 12345.678f) a.out[0] = 1; }
                else {
                  char* o = outp + (pix0 + (size_t)mb * a.W) * REC;
                  store8<T>(o, v);
                  store8<T>(o + a.out_plane_bytes, v + 8);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (EK == EK_SUB2 || EK == EK_PS2) {
      // activation (ReLU6 / LeakyReLU / PReLU / none), * alpha, [+ res1 at the output position], layout-aware store
      const float alpha = a.alpha;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int blk = grp * NB + nb, vblock = blk * 32;
        if (vblock >= a.cout_pad) continue;
        float slope_v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + NB * 32 + nb * 32 + 16 * (q >> 1) + 8 * lhe + 4 * (q & 1));
          slope_v[4 * q] = s4.x; slope_v[4 * q + 1] = s4.y; slope_v[4 * q + 2] = s4.z; slope_v[4 * q + 3] = s4.w;
        }
        const size_t sub = (size_t)lhe * HB;
        // PS2: block blk holds output row parity dy, output channels oc0 .. oc0+15; its first plane is sub-pixel dx = 0,
        // its second dx = 1 (virt_to_real_cout, pack.cpp).  SUB2: the block's two planes, as in the plain layout.
        const int cpb = EK == EK_PS2 ? (a.cout_real >> 2) / CW : 1;          // 16-channel planes of the shuffled output
        const int ps_dy = EK == EK_PS2 ? blk / cpb : 0;
        const int opl = EK == EK_PS2 ? blk - ps_dy * cpb : vblock / CW;
        char* outp = a.out + (size_t)(a.out_plane0 + opl) * a.out_plane_bytes + sub;
        const char* r1p = a.res1 ? a.res1 + (size_t)(a.r1_plane0 + opl) * a.r1_plane_bytes + sub : nullptr;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (EK == EK_SUB2 && (mb & 1)) continue;
          const int y = cur_y0 + wave * MB + mb;
          bool ok = y < a.H && xo < a.W;
          size_t opix0, ostep;   // first output record of this lane, distance (in records' planes) of the second one
          if constexpr (EK == EK_SUB2) {
            ok = ok && !(xo & 1);
            opix0 = ((size_t)cur_n * ((a.H + 1) >> 1) + (y >> 1)) * ((a.W + 1) >> 1) + (xo >> 1);
          } else {
            opix0 = ((size_t)cur_n * 2 * a.H + 2 * y + ps_dy) * (2 * (size_t)a.W) + 2 * xo;
          }
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float t = acc[nb][mb][i];
            if (a.act == ACT_RELU6) t = fminf(fmaxf(t, 0.f), 6.f);
            else { const float neg = t * slope_v[i]; t = t >= 0.f ? t : neg; }
            v[i] = t * alpha;
          }
          if (ok) {
            // the lane's two 8-channel groups: PS2 -> the same channels of two horizontally adjacent output pixels
            // (adjacent records of one plane); SUB2 -> the two planes of the block at one pixel
            char* o0 = outp + opix0 * REC;
            char* o1 = EK == EK_PS2 ? o0 + REC : o0 + a.out_plane_bytes;
            if (r1p) {
              float r[16];
              const char* q0 = r1p + opix0 * REC;
              load8<T>(q0, r); load8<T>(EK == EK_PS2 ? q0 + REC : q0 + a.r1_plane_bytes, r + 8);
#pragma unroll
              for (int i = 0; i < 16; ++i) v[i] += r[i];
            }
            store8<T>(o0, v);
            store8<T>(o1, v + 8);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (GEN) {
      // general epilogue, one 8-channel half (hq) of a 32-cout block at a time
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const int vbase = (grp * NB + nb) * 32 + 16 * hq + 8 * lhe;  // first of this lane's 8 channels
        if ((grp * NB + nb) * 32 >= a.cout_pad) continue;
        if (a.epi == EPI_NCHW_F32 && (grp * NB + nb) * 32 + 16 * hq >= a.cout_real) continue;  // nothing of this half is handed over
        float slope_v[8];
        if (a.act == ACT_PRELU) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + NB * 32 + nb * 32 + 16 * hq + 8 * lhe + 4 * q);
            slope_v[4 * q] = s4.x; slope_v[4 * q + 1] = s4.y; slope_v[4 * q + 2] = s4.z; slope_v[4 * q + 3] = s4.w;
          }
        }
        uint4 r1v[MB][RV8], r2v[MB][RV8];
        if (batch_res) {
          const int opl = vbase / CW;
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
            const int y = cur_y0 + wave * MB + mb;
            const bool ok = y < a.H && xo < a.W;
            const size_t ipix = ((size_t)cur_n * a.H + min(y, a.H - 1)) * a.W + min(xo, a.W - 1);
            const size_t orec = ipix * REC + (size_t)(vbase - opl * CW) * sizeof(T);
#pragma unroll
            for (int q = 0; q < RV8; ++q) {
              r1v[mb][q] = (a.res1 && ok) ? *reinterpret_cast<const uint4*>(a.res1 + (size_t)(a.r1_plane0 + opl) * a.r1_plane_bytes + orec + 16 * q)
                                          : make_uint4(0, 0, 0, 0);
              r2v[mb][q] = (a.res2 && ok) ? *reinterpret_cast<const uint4*>(a.res2 + (size_t)(a.r2_plane0 + opl) * a.r2_plane_bytes + orec + 16 * q)
                                          : make_uint4(0, 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int y = cur_y0 + wave * MB + mb;
          if (y < a.H && xo < a.W) {
            const size_t ipix = ((size_t)cur_n * a.H + y) * a.W + xo;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = acc[nb][mb][8 * hq + i];
            if (a.act == ACT_LRELU) {  // slope in [0,1] (checked on the host): max(v, slope*v)
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], v[i] * a.slope);
            } else if (a.act == ACT_PRELU) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float neg = v[i] * slope_v[i];  // unconditional: a select, not a branch per element
                v[i] = v[i] >= 0.f ? v[i] : neg;
              }
            } else if (a.act == ACT_RELU6) {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = fminf(fmaxf(v[i], 0.f), 6.f);
            }
            // where this 8-channel group lands
            const size_t opix = ipix; const int oc = vbase;   // (stride-2 / PixelShuffle layers have their own builds)
            {
              if (a.alpha != 1.f) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= a.alpha;
              }
              const int opl = oc / CW;
              const size_t orec = opix * REC + (size_t)(oc - opl * CW) * sizeof(T);
              if (batch_res) {
                if (a.res1) {
                  float r[8];
                  load8<T>(reinterpret_cast<const char*>(&r1v[mb][0]), r);
#pragma unroll
                  for (int i = 0; i < 8; ++i) v[i] += r[i];
                }
                if (a.res2) {
                  float r[8];
                  load8<T>(reinterpret_cast<const char*>(&r2v[mb][0]), r);
#pragma unroll
                  for (int i = 0; i < 8; ++i) v[i] = v[i] * a.gamma + r[i];
                }
              } else {
                if (a.res1 && (!a.bsvd_resid || vbase == 0)) {
                  float r[8];
                  const size_t rrec = (a.epi <= EPI_NHWC_PS2) ? orec : ipix * REC + (size_t)(vbase % CW) * sizeof(T);
                  load8<T>(a.res1 + (size_t)(a.r1_plane0 + (a.epi <= EPI_NHWC_PS2 ? opl : vbase / CW)) * a.r1_plane_bytes + rrec, r);
                  if (a.bsvd_resid) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) v[i] = r[i] - v[i];
                  } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += r[i];
                  }
                }
                if (a.res2) {
                  float r[8];
                  load8<T>(a.res2 + (size_t)(a.r2_plane0 + opl) * a.r2_plane_bytes + orec, r);
#pragma unroll
                  for (int i = 0; i < 8; ++i) v[i] = v[i] * a.gamma + r[i];
                }
              }
              if constexpr ((DBG & DBG_NO_STORE) != 0) {
                if (v[0] == 12345.678f) a.out[0] = 1;  // keep the values live without storing
              } else if (a.epi <= EPI_NHWC_PS2) {
                store8<T>(a.out + (size_t)(a.out_plane0 + opl) * a.out_plane_bytes + orec, v);
              } else {  // EPI_NCHW_F32
                float* o = reinterpret_cast<float*>(a.out);
                const size_t plane = (size_t)a.H * a.W;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                  if (vbase + i < a.cout_real)
                    o[((size_t)cur_n * a.cout_real + vbase + i) * plane + (size_t)y * a.W + xo] = v[i];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    { const unsigned long long t = stamp(); st_epi += t - st_t; st_t = t; }
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
  if constexpr ((DBG & DBG_STAMP) != 0) {
    const unsigned long long st_end = stamp();
    if (lane == 0 && a.dbg_buf) {
      unsigned long long* o = a.dbg_buf + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16;
      if (wave == 0) {
        o[0] = st_end - st_begin; o[1] = st_dma; o[2] = st_mma; o[3] = st_epi; o[4] = st_bar; o[5] = kt + 1;
        o[6] = st_pro; o[7] = st_store; o[8] = st_wait; o[9] = __builtin_amdgcn_s_memrealtime() - st_rt0;
        o[14] = st_rt0; o[15] = __builtin_amdgcn_s_memrealtime();
      }
      if (wave == NW - 1) { o[10] = st_mma; o[11] = st_bar; o[12] = st_wait; o[13] = st_epi; }
    }
  }
}

template <typename T, int NB, int MB, int NW, int DBG, int EK>
static void launch_t(ss4k_ctx* ctx, const ConvArgs& a0, int groups, hipStream_t st) {
  using G = Geo<T, MB, NW>;
  constexpr size_t lds = lds_bytes<T, NB, MB, NW>();
  constexpr int per_cu = wgs_per_cu<T, NB, MB, NW>();
  static_assert(lds * per_cu <= 160 * 1024, "LDS budget");
  ConvArgs a = a0;
  a.tiles_y = (a.H + G::TH - 1) / G::TH;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const void* fn = reinterpret_cast<const void*>(&conv3x3_kernel<T, NB, MB, NW, DBG, EK>);
  if (ctx->lds_attr_set.insert(fn).second)  // per context = per device
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * per_cu / groups * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  ctx->prof_family = std::is_same<T, float>::value
      ? (NB == 1 ? "conv3x3_kernel<float,1> (32-cout tile, exact fp32 MFMA 32x32x2)" : "conv3x3_kernel<float,2> (64-cout tile, exact fp32 MFMA 32x32x2)")
      : (NB == 1 ? (MB == 5 ? "conv3x3_kernel<__half,1,5> (32-cout tile, 20 rows, 32x32x16 MFMA)" : "conv3x3_kernel<__half,1,4> (32-cout tile, 16 rows, 32x32x16 MFMA)")
                 : "conv3x3_kernel<__half,2,4> (64-cout tile, 32x32x16 MFMA)");
  hipLaunchKernelGGL((conv3x3_kernel<T, NB, MB, NW, DBG, EK>), dim3(gx, groups), dim3(64 * NW), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

#ifdef SS4K_DEV
// ss4k_bench_conv only: instrumented (phase stamps) and alternative tile shapes of the fp16 kernel
template <int NB, int MB, int NW>
static void launch_dbg(ss4k_ctx* ctx, const ConvArgs& a, int groups, hipStream_t st) {
  switch (a.dbg & 0xff) {
    case DBG_STAMP: launch_t<__half, NB, MB, NW, DBG_STAMP, EK_PLAIN>(ctx, a, groups, st); break;
    // timing-only ablations of the memory traffic (results are garbage), see profiles/NOTES_r01_r03.md 4.1:
    //   33: halo tiles from a 2 MB L2-resident window, no output stores   (no fabric traffic)
    //   49: real halo tiles, no output stores                             (no fabric writes)
    //   48: halo tiles from the 2 MB window, real stores                  (no fabric reads)
    //   34: every DMA instruction reads one hot cache line                (no L2 traffic either)
    case DBG_STAMP | DBG_NO_STORE: launch_t<__half, NB, MB, NW, DBG_STAMP | DBG_NO_STORE, EK_PLAIN>(ctx, a, groups, st); break;
    case DBG_STAMP | DBG_NO_STORE | DBG_NO_EPILOGUE: launch_t<__half, NB, MB, NW, DBG_STAMP | DBG_NO_STORE | DBG_NO_EPILOGUE, EK_PLAIN>(ctx, a, groups, st); break;
    case DBG_STAMP | DBG_NO_EPILOGUE: launch_t<__half, NB, MB, NW, DBG_STAMP | DBG_NO_EPILOGUE, EK_PLAIN>(ctx, a, groups, st); break;
    case DBG_STAMP | DBG_NO_MMA: launch_t<__half, NB, MB, NW, DBG_STAMP | DBG_NO_MMA, EK_PLAIN>(ctx, a, groups, st); break;
    case 0: launch_t<__half, NB, MB, NW, 0, EK_PLAIN>(ctx, a, groups, st); break;
    default: throw Error(SS4K_EINVAL, "ss4k_bench_conv: flags are 0 or 32 (phase stamps), | tile-shape id << 8");
  }
}

#endif  // SS4K_DEV

void launch_conv3x3(ss4k_ctx* ctx, const ConvArgs& a0, int dtype, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.zero_page = ctx->zero_page();
#ifdef SS4K_DEV
  const bool split64 = a.cout_pad == -64;   // dev experiment (models.cpp SS4K_SPLIT64): 64 couts as two 32-cout groups
  if (split64) a.cout_pad = 64;
#else
  constexpr bool split64 = false;
#endif
  const int nb = (a.cout_pad <= 32 || split64) ? 1 : 2;
  const int groups = (a.cout_pad + nb * 32 - 1) / (nb * 32);
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "conv3x3: empty grid");
  SS4K_REQUIRE(a.act != ACT_LRELU || (a.slope >= 0.f && a.slope <= 1.f), "conv3x3: LeakyReLU slope must be in [0,1]");
  SS4K_REQUIRE(!a.ups2 || ((a.H % 2 == 0) && (a.W % 2 == 0)), "conv3x3: ups2 needs even grid");
  SS4K_REQUIRE((double)a.N * a.H * a.W < 2147483648.0, "conv3x3: a plane holds at most 2^31 pixels");
  ProfScope prof(ctx, st, PROF_CONV);
  // fp16 tile shapes <couts/32, rows per wave, waves>; see DESIGN.md 4.3 and profiles/NOTES_r01_r03.md 4.1 for how they were chosen
#ifdef SS4K_DEV
  // dev library: fp16 layers of a supported shape with a plain epilogue on the register-stationary kernel (conv_rs.hip)
  if (a.wrs && dtype == SS4K_F16 && !a.dbg && a.epi == EPI_NHWC && !a.bsvd_resid && a.act != ACT_RELU6) {
    launch_conv3x3_rs(ctx, a, st);
  } else
#endif
  if (nb == 1 && conv3x3_w16n_eligible(a, dtype)) {
    launch_conv3x3_w16n(ctx, a, st);   // <= 4 output channels, NCHW fp32 hand-off: one 16-cout block on the 16x16x32 MFMA (conv_w16n.hip)
  } else if (a.wide && nb == 2 && conv3x3_w16_eligible(a, dtype)) {
    launch_conv3x3_w16(ctx, a, st);    // the same tile on the 16x16x32 MFMA (conv_w16.hip; the layer has a w16 blob)
  } else if (a.wide && nb == 2 && conv3x3_wide_eligible(a, dtype)) {
    launch_conv3x3_wide(ctx, a, st);   // 64-cout groups, plain epilogue: conv_dense.hip's single-layer build (bit-identical to <__half,2,4,4>)
  } else
#ifdef SS4K_DEV
  if (a.dbg) {
    SS4K_REQUIRE(dtype == SS4K_F16, "instrumented builds exist for fp16 only");
    const int shape = (a.dbg >> 8) & 7;
    if (nb == 1) {
      switch (shape) {
        case 1: launch_dbg<1, 2, 4>(ctx, a, groups, st); break;
        case 3: launch_dbg<1, 2, 8>(ctx, a, groups, st); break;
        case 4: launch_dbg<1, 4, 8>(ctx, a, groups, st); break;
        default: launch_dbg<1, 4, 4>(ctx, a, groups, st); break;
      }
    } else {
      switch (shape) {
        case 2: launch_dbg<2, 4, 4>(ctx, a, groups, st); break;
        case 3: launch_dbg<2, 4, 8>(ctx, a, groups, st); break;
        case 4: launch_dbg<2, 2, 4>(ctx, a, groups, st); break;
        default: launch_dbg<2, 2, 8>(ctx, a, groups, st); break;
      }
    }
  } else
#endif
  {
    SS4K_REQUIRE(a.dbg == 0, "instrumented conv builds live in libss4k_hip_dev.so only");
    const int ek = a.epi == EPI_NHWC_SUB2 ? EK_SUB2 : a.epi == EPI_NHWC_PS2 ? EK_PS2 : (a.epi == EPI_NHWC && !a.bsvd_resid) ? EK_PLAIN : EK_GENERAL;
    SS4K_REQUIRE(ek == EK_PLAIN || ek == EK_GENERAL || (!a.res2 && !a.bsvd_resid), "conv3x3: stride-2 / PixelShuffle epilogues take res1 only");
    SS4K_REQUIRE(ek != EK_SUB2 || !a.res1, "conv3x3: the stride-2 epilogue takes no residual");
    SS4K_REQUIRE(ek != EK_PS2 || ((a.cout_real & 63) == 0 && a.cout_pad == a.cout_real), "conv3x3: PixelShuffle(2) needs a multiple of 64 output channels");
#define SS4K_LAUNCH_EK(T_, NB_)                                                          \
    switch (ek) {                                                                        \
      case EK_PLAIN: launch_t<T_, NB_, 4, 4, 0, EK_PLAIN>(ctx, a, groups, st); break;     \
      case EK_SUB2: launch_t<T_, NB_, 4, 4, 0, EK_SUB2>(ctx, a, groups, st); break;       \
      case EK_PS2: launch_t<T_, NB_, 4, 4, 0, EK_PS2>(ctx, a, groups, st); break;         \
      default: launch_t<T_, NB_, 4, 4, 0, EK_GENERAL>(ctx, a, groups, st); break;         \
    }
    // 32-cout body layers: 20-row tiles (five rows per wave) where they cut the image's rows with less waste than 16-row
    // ones (360 rows = 18 x 20 but 22.5 x 16: the last 16-row tile row computes 8 rows of nothing; the halo overhead of a
    // stage is 1.17 instead of 1.20 too) AND there is still at least one tile in flight per workgroup slot (a 1-frame 720p job
    // on one chain has 360 such tiles for 512 slots: measured -1.3 %; 4- and 2-frame jobs: +1.7 % / +2.2 % on the whole network).
    // Same kernel, one more build, bit-identical results; SS4K_MB=4/5 (read when a model is built) is the A/B switch (tools/env_ab.py).  24-row tiles
    // (six rows per wave) were built and measured too: +0.2 % over 20-row ones at 4 frames, -1.8 % at 2: not kept; 40-row
    // tiles on eight waves (one workgroup per CU, 17 % fewer L2->LDS bytes): +0.3 %; 8-row tiles at three workgroups per
    // CU: -5.3 % (-5.9 % on 1-frame jobs).
    const auto waste = [&](int th) { return (double)((a.H + th - 1) / th * th) / a.H; };
    // tiles in flight: a frame lane's launch (grid_share set) shares the chip with the other chain's launch
    const long long tiles20 = (long long)a.N * ((a.H + 19) / 20) * a.tiles_x * (a.grid_share > 0.f ? 2 : 1);
    const bool mb5 = dtype == SS4K_F16 && nb == 1 && ek == EK_PLAIN &&
                     (a.mb_override ? a.mb_override == 5 : (waste(20) < waste(16) - 1e-9 && tiles20 >= 2LL * ctx->num_cu));
#ifdef SS4K_DEV
    // experiment (SS4K_S3=1, dev library): three-stage ring of halo tiles, conv_s3.hip
    if (a.s3 && conv3x3_s3_eligible(a, dtype)) launch_conv3x3_s3(ctx, a, st);
    else
    // experiment (SS4K_MB=3, dev library): 12-row tiles at THREE workgroups per CU for the 32-cout layers
    if (a.mb_override == 3 && dtype == SS4K_F16 && nb == 1 && ek == EK_PLAIN) launch_t<__half, 1, 3, 4, 0, EK_PLAIN>(ctx, a, groups, st);
    else
#endif
    if (mb5) launch_t<__half, 1, 5, 4, 0, EK_PLAIN>(ctx, a, groups, st);
    else if (dtype == SS4K_F16) { if (nb == 1) { SS4K_LAUNCH_EK(__half, 1) } else { SS4K_LAUNCH_EK(__half, 2) } }
    else { if (nb == 1) { SS4K_LAUNCH_EK(float, 1) } else { SS4K_LAUNCH_EK(float, 2) } }
#undef SS4K_LAUNCH_EK
  }
  prof.done(a0.flops);
}

}  // namespace ss4k
