// Dense-block layer pair: conv_k and conv_{k+1} of a Residual Dense Block as ONE launch (RRDBNet: (conv1, conv2) and (conv3, conv4) of
// every RDB; realesrgan/factory.py:112-127 builds them one nn.Conv2d at a time, the reference's engine fuses the graph, :206-230).
//
// Why.  conv_{k+1} reads exactly conv_k's input planes plus conv_k's output: launched one after the other the two layers stream the
// same planes twice and the 32-channel x_k makes a round trip through memory in between.  Here a workgroup
//   * streams the shared input planes ONCE: per K-chunk one halo tile feeds two accumulator sets - conv_k on the output tile grown by
//     one pixel (18 rows x 32 columns) and conv_{k+1}'s partial sum on the tile itself - and every activation fragment read from LDS
//     feeds up to six MFMAs (three of each layer) instead of three: the x-phase IS the 64-cout tile body of conv_mfma.hip with
//     [W_k ; W_{k+1}] as its two cout blocks;
//   * activates x_k, rounds it to fp16 exactly as the store does, keeps it in LDS (positions outside the image are ZEROS: they are
//     conv_{k+1}'s padding, not conv_k evaluated on padded input), writes its interior to memory (conv5 and the later layers read it)
//     and finishes conv_{k+1} with two K-chunks from LDS.
// Per pixel and RDB the planes read drop from 4+6+8+10 = 28 to 4+8 = 12 (conv5's 12 stay), the launches from 5 to 3.
//
// Geometry (fp16, 4 waves, TWO workgroups per CU like the kernel it replaces):
//   output tile of conv_{k+1}: 16 rows x 30 columns (lane l of an MFMA's 32 pixels <-> image column x0 - 1 + l; lanes 1..30 are stored)
//   x_k tile:                  18 rows x 32 columns (rows y0 - 1 .. y0 + 16, lanes 0..31), LDS image 18 x 34 pixels (column = lane + 1)
//   input halo tile:           20 rows x 34 columns per K-chunk (rows y0 - 2 .., columns x0 - 2 ..)
//   wave w: conv_{k+1} rows 4w .. 4w+3 and conv_k rows y0 + 4w .. y0 + 4w + 3 - the SAME six input rows, so the two layers share every
//   fragment; x_k's two extra rows (y0 - 1: wave 0, y0 + 16: wave 3) cost those waves one more row of conv_k MFMAs.
//   MFMA work: x 32/30 on every layer (two of 32 lanes are halo), x 5/4 on conv_k for waves 0 and 3.
// LDS (80.6 KB, two workgroups per CU): two 21.3 KB halo-tile buffers, two 18 KB weight stages (a chunk's W_k and W_{k+1} fragments, each
// 9 KB, DMA'd from the two layers' own packed blobs - pack.cpp, nothing is repacked), 256 B of biases.  After the last input chunk the two
// tile buffers are free: x_k's plane 0 lives in buffer 0, plane 1 in buffer 1 (19.1 KB each), and the next tile's first chunk is
// prefetched into buffer 0 while conv_{k+1}'s last chunk reads buffer 1.
//
// Results are BIT-IDENTICAL to the two launches on the LDS-weights kernel (tests/test_gpu_dense_pair.py): same packed fragments, fp32
// accumulators start from the bias, MFMAs per output in (K-chunk, dx, dy) order, LeakyReLU as max(t, slope t), round-to-nearest fp16.
#include "common.h"
#include "conv_tile.h"

namespace ss4k {
namespace dense {

constexpr int NW = 4, MB = 4, TH = NW * MB, TWO = TW - 2;
constexpr int XH = TH + 4;                          // rows of an input halo tile
constexpr int REC = 32, SPR = 2;
constexpr int ROWB = IN_W * REC;                    // bytes of one 34-pixel row of an LDS image
constexpr int XT_SLOTS = XH * IN_W * SPR;           // 1360 16-byte slots
constexpr int XT_BYTES = XT_SLOTS * 16;             // 21760
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 22 wave-level DMA instructions
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 6
constexpr int WC = 9 * 64 * 16;                     // one layer's fragments for one K-chunk: 9216 bytes
constexpr int W_BYTES = 2 * WC;                     // a weight stage
constexpr int NDMA_W = (18 + NW - 1) / NW;          // 5
constexpr int NDMA = DMA_PER_WAVE + NDMA_W;         // 11 DMA slots per wave and chunk
constexpr size_t LDS_BYTES = 2 * XT_BYTES + 2 * W_BYTES + 64 * 4;
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert((TH + 2) * ROWB <= XT_BYTES, "x_k's plane image fits a tile buffer");

// LeakyReLU with a slope in [0, 1] as max(t, slope t) - the value conv_mfma.hip's epilogue computes with fmaxf - in ONE instruction:
// v_med3_f32(t, slope t, +inf).  (fmaxf costs two: in IEEE mode the compiler first quiets a possible signalling NaN with v_max(t, t).)
__device__ __forceinline__ float lrelu(float t, float slope) { return __builtin_amdgcn_fmed3f(t, t * slope, __builtin_inff()); }

// K1T: K-chunks of conv_k when known at compile time (RRDBNet: 4 and 8), 0 = read from the arguments
//   STAMP (dev library): per-wave cycle totals of the tile's phases (s_memtime), see launch_conv3x3_dense2
template <int K1T, bool STAMP = false>
__global__ __launch_bounds__(64 * NW, 2) void conv3x3_dense2_kernel(const DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int K1 = K1T ? K1T : a.nchunks0 + a.nchunks1;   // K-chunks of conv_k = the planes both layers read (even, host-checked)
  // STAMP: [0] tile setup + accumulator init, [1] x-phase reads + MFMAs + DMA issue, [2] x-phase vmcnt wait, [3] x-phase barrier, [4] x_k
  // epilogue, [5] barrier after it, [6] x_k-phase reads + MFMAs, [7] barrier between its chunks, [8] hand-over wait, [9] hand-over barrier,
  // [10] conv_{k+1} epilogue
  unsigned long long ph[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0, rt0 = 0, ct0 = 0;
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
  const int ntiles = a.N * a.tiles_y * a.tiles_x;

  // XCD-banded persistent tile walk (conv_mfma.hip): placement only, never results
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
  // operand read base: LDS image row 4w + 1, column lr + dx.  Input tile: the wave's six rows are 4w+1 .. 4w+6 (+ ir * ROWB);
  // x_k image: rows 4w .. 4w+5 ((ir - 1) * ROWB).
  int rd_base[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = lr + dx;
    rd_base[dx] = (((wave * MB + 1) * IN_W + x) * SPR + (lh ^ swz(x))) * 16;
  }
  // per-lane source of every halo-tile DMA instruction of this wave: byte offset inside a plane (32 bits: the host sends bigger
  // planes down the two-launch route), OOB = the zero page.  LDS slot s = 64k + lane holds pixel (row, x) = (s / 2) divmod 34, source
  // channel group (s & 1) ^ swz(x).  Recomputed per tile (six divisions by a constant) instead of kept in registers.
  uint32_t src_off[DMA_PER_WAVE];
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TWO;
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
      const int s = (wave + NW * j) * 64 + lane;
      const int p = s >> 1, gq = s & 1;
      const int row = p / IN_W, x = p - row * IN_W;
      const int iy = y0 - 2 + row, ix = x0 - 2 + x;
      const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      src_off[j] = ok ? ((uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix) * REC + (uint32_t)((gq ^ swz(x)) * 16) : OOB;
    }
  };

  // a prefetch = an optional halo tile (DMA_PER_WAVE instructions per wave) + an optional weight stage (18 KB = two 9 KB pieces)
  const char* pf_plane = nullptr; const char* pf_wa = nullptr; const char* pf_wb = nullptr;
  uint32_t pf_tdst = 0, pf_wdst = 0; bool pf_tile = false, pf_w = false;
  auto plane_of = [&](int c) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  auto dma_op = [&](int idx) {
    if (idx < DMA_PER_WAVE) {
      const int k = wave + NW * idx;
      if (pf_tile && k < XT_DMA) {
        const char* src = src_off[idx] != OOB ? pf_plane + src_off[idx] : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(pf_tdst + k * 1024);
        if (k * 64 + lane < XT_SLOTS) dma16(src, dst);   // lanes past the tile's last slot are masked off (EXEC)
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - DMA_PER_WAVE);
      if (pf_w && k < 18) dma16((k < 9 ? pf_wa + k * 1024 : pf_wb + (k - 9) * 1024) + lane * 16, __builtin_amdgcn_readfirstlane(pf_wdst + k * 1024));
    }
  };

  float* bias_lds = reinterpret_cast<float*>(smem + 2 * XT_BYTES + 2 * W_BYTES);  // [conv_k 32][conv_{k+1} 32]
  if (tid < 64) bias_lds[tid] = tid < 32 ? a.bias1[tid] : a.bias2[tid - 32];

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  int wbuf = 0;                                      // weight stage of the tile's first chunk (the tile buffer is always 0)
  pf_tile = true; pf_w = true; pf_plane = plane_of(0); pf_tdst = lds0; pf_wa = a.w1; pf_wb = a.w2; pf_wdst = lds0 + 2 * XT_BYTES;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  dma_wait();
  __syncthreads();

  const float slope = a.slope;
  while (true) {
    f32x16 acc1[MB], acc1x, acc2[MB];
    {
      float b1[16], b2[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 u = *reinterpret_cast<const float4*>(bias_lds + 16 * (q >> 1) + 8 * lh + 4 * (q & 1));
        const float4 v = *reinterpret_cast<const float4*>(bias_lds + 32 + 16 * (q >> 1) + 8 * lh + 4 * (q & 1));
        b1[4 * q] = u.x; b1[4 * q + 1] = u.y; b1[4 * q + 2] = u.z; b1[4 * q + 3] = u.w;
        b2[4 * q] = v.x; b2[4 * q + 1] = v.y; b2[4 * q + 2] = v.z; b2[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc1x[i] = b1[i];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { acc1[mb][i] = b1[i]; acc2[mb][i] = b2[i]; }
      }
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    int tb = 0;
    stamp(0);

    // ------------------------------------------------ x-phase: the planes both layers read, one halo tile per K-chunk
    for (int c = 0; c < K1; ++c) {
      if (c + 1 < K1) {
        pf_tile = true; pf_w = true; pf_plane = plane_of(c + 1); pf_tdst = lds0 + (tb ^ 1) * XT_BYTES;
        pf_wa = a.w1 + (size_t)(c + 1) * WC; pf_wb = a.w2 + (size_t)(c + 1) * WC;
      } else {   // conv_{k+1}'s two x_k chunks: weights only (chunks K1, K1 + 1 of its blob are contiguous)
        pf_tile = false; pf_w = true; pf_wa = a.w2 + (size_t)K1 * WC; pf_wb = pf_wa + WC;
      }
      pf_wdst = lds0 + 2 * XT_BYTES + (wbuf ^ 1) * W_BYTES;
      const char* tbp = smem + tb * XT_BYTES;
      const char* wbp = smem + 2 * XT_BYTES + wbuf * W_BYTES + lane * 16;
      // Fragments.  An input row's fragment is used by ONE step of the (dx, row) walk only (by the <= 3 MFMAs per layer that read row ir
      // at tap column dx), so three rotating registers hold the rows of steps t, t + 1, t + 2 and step t + 3 is loaded as soon as
      // step t's MFMAs have issued (two steps of latency cover at a quarter of the registers six live rows would take).  The weight
      // fragments of a tap column stay for its six steps and are refilled in place for the next column after their last MFMA.
      uint4 wf1[3], wf2[3], af[3], afx;
      auto af_load = [&](int t) { return *reinterpret_cast<const uint4*>(tbp + rd_base[t / (MB + 2)] + (t % (MB + 2)) * ROWB); };
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        wf1[dy] = *reinterpret_cast<const uint4*>(wbp + (dy * 64) * 16);
        wf2[dy] = *reinterpret_cast<const uint4*>(wbp + WC + (dy * 64) * 16);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = af_load(t);
      // the extra conv_k row: wave 0 -> x_k row 0 reads input rows 0 (this fragment), 1, 2; wave 3 -> x_k row 17 reads input rows 17, 18
      // and 19 (this fragment)
      if (wave == 0) afx = *reinterpret_cast<const uint4*>(tbp + rd_base[0] - ROWB);
      if (wave == NW - 1) afx = *reinterpret_cast<const uint4*>(tbp + rd_base[0] + (MB + 2) * ROWB);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool more = g + 1 < 3;
        int m = 0;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir) {
          const int t = g * (MB + 2) + ir;
          const uint4& cur = af[t % 3];
          if (ir == 0 && wave == 0) {
            acc1x = mma<__half>(wf1[0], afx, acc1x);
            if (more) afx = *reinterpret_cast<const uint4*>(tbp + rd_base[g + 1] - ROWB);
          }
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int mb = ir - dy;
            if (mb >= 0 && mb < MB) {
              acc1[mb] = mma<__half>(wf1[dy], cur, acc1[mb]);
              acc2[mb] = mma<__half>(wf2[dy], cur, acc2[mb]);
              if (m % 3 == 1 && g * 4 + m / 3 < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(g * 4 + m / 3);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
          if (ir == 0 && wave == 0) acc1x = mma<__half>(wf1[1], cur, acc1x);
          if (ir == 1 && wave == 0) acc1x = mma<__half>(wf1[2], cur, acc1x);
          if (ir == MB && wave == NW - 1) acc1x = mma<__half>(wf1[0], cur, acc1x);
          if (ir == MB + 1 && wave == NW - 1) {
            acc1x = mma<__half>(wf1[1], cur, acc1x);
            acc1x = mma<__half>(wf1[2], afx, acc1x);
            if (more) afx = *reinterpret_cast<const uint4*>(tbp + rd_base[g + 1] + (MB + 2) * ROWB);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 3 < 3 * (MB + 2)) af[t % 3] = af_load(t + 3);
          if (more) {   // weights of the next tap column, in place, as soon as a register's last MFMA of this column has issued
            if (ir >= MB - 1) wf2[ir - (MB - 1)] = *reinterpret_cast<const uint4*>(wbp + WC + (((g + 1) * 3 + ir - (MB - 1)) * 64) * 16);
            if (ir == MB) wf1[0] = *reinterpret_cast<const uint4*>(wbp + (((g + 1) * 3 + 0) * 64) * 16);
            if (ir == MB + 1) {
              wf1[1] = *reinterpret_cast<const uint4*>(wbp + (((g + 1) * 3 + 1) * 64) * 16);
              wf1[2] = *reinterpret_cast<const uint4*>(wbp + (((g + 1) * 3 + 2) * 64) * 16);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      stamp(1);
      dma_wait();        // the next chunk's tile / weights have landed
      stamp(2);
      __syncthreads();   // and every wave is done with this chunk's buffers
      stamp(3);
      tb ^= 1; wbuf ^= 1;
    }
    // here: tb == 0 (K1 is even), weight stage wbuf holds conv_{k+1}'s chunks K1 (first half) and K1 + 1 (second half); both tile
    // buffers are free

    // ------------------------------------------------ x_k: activation, fp16, -> LDS (zeros outside the image) and -> memory (interior)
    {
      const int xcol = cur_x0 - 1 + lr;
      const bool col_in = xcol >= 0 && xcol < a.W;
      const bool col_st = col_in && lr >= 1 && lr <= TWO;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));   // re-derive the store pointers per tile (hoisted they cost registers through the MFMA loops)
      char* o1 = a.out1 + (size_t)a.out1_plane0 * a.out1_plane_bytes + (size_t)lhe * 16;
      const uint32_t lrow = (uint32_t)(((lr + 1) * SPR + (lhe ^ swz(lr + 1))) * 16);
      auto put = [&](const f32x16& acc, int j, bool to_mem) {
        const int y = cur_y0 - 1 + j;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float t = acc[i]; v[i] = lrelu(t, slope); }
        uint4 h0, h1;
        __half* p0 = reinterpret_cast<__half*>(&h0);
        __half* p1 = reinterpret_cast<__half*>(&h1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { p0[i] = __float2half(v[i]); p1[i] = __float2half(v[8 + i]); }
        const uint32_t msk = (col_in && y >= 0 && y < a.H) ? 0xFFFFFFFFu : 0u;   // outside the image: zeros (conv_{k+1}'s padding)
        char* l = smem + j * ROWB + lrow;
        *reinterpret_cast<uint4*>(l) = make_uint4(h0.x & msk, h0.y & msk, h0.z & msk, h0.w & msk);
        *reinterpret_cast<uint4*>(l + XT_BYTES) = make_uint4(h1.x & msk, h1.y & msk, h1.z & msk, h1.w & msk);
        if (to_mem && col_st && y < a.H) {
          char* o = o1 + ((size_t)(cur_n * a.H + y) * a.W + xcol) * REC;
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&h0), reinterpret_cast<u32x4*>(o));
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&h1), reinterpret_cast<u32x4*>(o + a.out1_plane_bytes));
        }
      };
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) put(acc1[mb], wave * MB + 1 + mb, true);
      if (wave == 0) put(acc1x, 0, false);
      if (wave == NW - 1) put(acc1x, TH + 1, false);
    }
    stamp(4);
    lds_barrier();   // x_k is visible; (its stores to memory stay in flight)
    stamp(5);

    // ------------------------------------------------ conv_{k+1}'s last two K-chunks: x_k from LDS
    if (next_tile >= 0) setup_tile(next_tile, n, y0, x0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // chunk K1: the next tile's first weight stage goes into the free one; chunk K1 + 1: its first halo tile into buffer 0
      pf_tile = h == 1 && next_tile >= 0; pf_w = h == 0 && next_tile >= 0;
      pf_plane = plane_of(0); pf_tdst = lds0; pf_wa = a.w1; pf_wb = a.w2; pf_wdst = lds0 + 2 * XT_BYTES + (wbuf ^ 1) * W_BYTES;
      const char* tbp = smem + h * XT_BYTES - ROWB;       // image row 4w + ir = read base row (4w + 1) + (ir - 1)
      const char* wbp = smem + 2 * XT_BYTES + wbuf * W_BYTES + h * WC + lane * 16;
      uint4 wf[3], af[3];
      auto af_load = [&](int t) { return *reinterpret_cast<const uint4*>(tbp + rd_base[t / (MB + 2)] + (t % (MB + 2)) * ROWB); };
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) wf[dy] = *reinterpret_cast<const uint4*>(wbp + (dy * 64) * 16);
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = af_load(t);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool more = g + 1 < 3;
        int m = 0;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir) {
          const int t = g * (MB + 2) + ir;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int mb = ir - dy;
            if (mb >= 0 && mb < MB) {
              acc2[mb] = mma<__half>(wf[dy], af[t % 3], acc2[mb]);
              if (m % 3 == 1 && g * 4 + m / 3 < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(g * 4 + m / 3);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 3 < 3 * (MB + 2)) af[t % 3] = af_load(t + 3);
          if (more && ir >= MB - 1) wf[ir - (MB - 1)] = *reinterpret_cast<const uint4*>(wbp + (((g + 1) * 3 + ir - (MB - 1)) * 64) * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      stamp(6);
      if (h == 0) { lds_barrier(); stamp(7); }   // every wave is done with plane 0's image: the next tile's halo tile may land there
    }
    // hand the buffers to the next tile BEFORE the epilogue (conv_mfma.hip): the stores drain under its first chunk's MFMAs
    if (next_tile >= 0) {
      dma_wait();
      stamp(8);
      __syncthreads();
      stamp(9);
      wbuf ^= 1;
    }

    // ------------------------------------------------ conv_{k+1}'s epilogue: activation, fp16, interior lanes -> memory
    {
      const int xcol = cur_x0 - 1 + lr;
      const bool col_st = xcol < a.W && lr >= 1 && lr <= TWO;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));
      char* o2 = a.out2 + (size_t)a.out2_plane0 * a.out2_plane_bytes + (size_t)lhe * 16;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int y = cur_y0 + wave * MB + mb;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = lrelu(acc2[mb][i], slope);
        if (col_st && y < a.H) {
          char* o = o2 + ((size_t)(cur_n * a.H + y) * a.W + xcol) * REC;
          store8<__half>(o, v);
          store8<__half>(o + a.out2_plane_bytes, v + 8);
        }
      }
    }
    stamp(10);
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg_buf && blockIdx.x < 1024) {
      unsigned long long* o = a.dbg_buf + ((size_t)blockIdx.x * 4 + wave) * 16;
      for (int k = 0; k < 11; ++k) o[k] = ph[k];
      o[11] = (unsigned long long)(kt + 1);
      o[12] = ((__builtin_amdgcn_s_memtime() - ct0) << 20) / (__builtin_amdgcn_s_memrealtime() - rt0 + 1);   // shader cycles per 100 MHz tick, x 2^20
    }
  }
}

}  // namespace dense

bool conv3x3_dense2_eligible(int nchunks_a, int cout_pad_a, int nchunks_b, int cout_pad_b) {
  return cout_pad_a == 32 && cout_pad_b == 32 && nchunks_b == nchunks_a + 2 && nchunks_a >= 2 && nchunks_a % 2 == 0;
}

void launch_conv3x3_dense2(ss4k_ctx* ctx, const DenseArgs& a0, hipStream_t st) {
  using namespace dense;
  DenseArgs a = a0;
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "dense pair: empty grid");
  SS4K_REQUIRE((a.nchunks0 + a.nchunks1) % 2 == 0 && a.nchunks0 + a.nchunks1 >= 2, "dense pair: conv_k needs an even number of K-chunks");
  SS4K_REQUIRE(a.slope >= 0.f && a.slope <= 1.f, "dense pair: LeakyReLU slope must be in [0,1]");
  SS4K_REQUIRE((double)a.N * a.H * a.W < 2147483648.0, "dense pair: a plane holds at most 2^31 pixels");
  a.tiles_x = (a.W + TWO - 1) / TWO; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 * (a.grid_share > 0.f ? a.grid_share : 1.f))));
#ifdef SS4K_DEV
  if (const char* e = std::getenv("SS4K_DENSE_GRID")) gx = std::min(ntiles, std::max(1, std::atoi(e)));   // e.g. 256: one workgroup per CU (stamps without a partner)
#endif
  const ProfEvent pe = ctx->prof_begin(st, PROF_CONV);
  auto go = [&](auto kern) {
    const void* fn = reinterpret_cast<const void*>(kern);
    if (ctx->lds_attr_set.insert(fn).second)
      SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * NW), LDS_BYTES, st, a);
  };
#ifdef SS4K_DEV
  static const bool stamp_mode = std::getenv("SS4K_DENSE_STAMP") && std::getenv("SS4K_DENSE_STAMP")[0] == '1';
  if (stamp_mode) {   // phase cycle counters of every wave of the first 1024 workgroups
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) SS4K_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 4 * 16 * 8));
    SS4K_HIP(hipMemsetAsync(dbuf, 0, 1024 * 4 * 16 * 8, st));
    a.dbg_buf = dbuf;
    go(&conv3x3_dense2_kernel<0, true>);
    SS4K_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hb(1024 * 4 * 16);
    SS4K_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
    static int printed = 0;
    if (printed++ < 6) {
      for (int w = 0; w < 4; ++w) {
        double acc[11] = {0}, tiles = 0, clk = 0; int nw = 0;
        for (int wg = 0; wg < 1024; ++wg) {
          const unsigned long long* o = &hb[((size_t)wg * 4 + w) * 16];
          if (!o[11]) continue;
          for (int k = 0; k < 11; ++k) acc[k] += (double)o[k];
          tiles += (double)o[11]; clk += (double)o[12] / 1048576.0 * 100.0; ++nw;
        }
        if (tiles > 0) {
          double tot = 0; for (double v : acc) tot += v;
          std::fprintf(stderr, "[dense K1=%d wave %d] %.0f MHz, cycles per tile: total %.0f | setup %.0f | x: mfma %.0f wait %.0f barrier %.0f | x_k epilogue %.0f barrier %.0f | x_k chunks %.0f barrier %.0f | hand-over wait %.0f barrier %.0f | epilogue %.0f\n",
                       a.nchunks0 + a.nchunks1, w, clk / nw, tot / tiles, acc[0] / tiles, acc[1] / tiles, acc[2] / tiles, acc[3] / tiles, acc[4] / tiles, acc[5] / tiles,
                       acc[6] / tiles, acc[7] / tiles, acc[8] / tiles, acc[9] / tiles, acc[10] / tiles);
        }
      }
    }
    ctx->prof_end(pe, st, a0.flops);
    return;
  }
#endif
  switch (a.nchunks0 + a.nchunks1) {
    case 4: go(&conv3x3_dense2_kernel<4>); break;
    case 8: go(&conv3x3_dense2_kernel<8>); break;
    default: go(&conv3x3_dense2_kernel<0>); break;
  }
  SS4K_HIP(hipGetLastError());
  ctx->prof_end(pe, st, a0.flops);
}

}  // namespace ss4k
