// Dense-block layer pair: conv_k and conv_{k+1} of a Residual Dense Block as ONE launch (RRDBNet: (conv1, conv2) and (conv3, conv4) of
// every RDB; realesrgan/factory.py:112-127 builds them one nn.Conv2d at a time, the reference's engine fuses the graph, :206-230).
//
// Why.  conv_{k+1} reads exactly conv_k's input planes plus conv_k's output: launched one after the other the two layers stream the
// same planes twice and the 32-channel x_k makes a round trip through memory in between.  Here a workgroup
//   * streams the shared input planes ONCE: per K-chunk one halo tile feeds two accumulator sets - conv_k and conv_{k+1}'s partial sum
//     on the same 16 x 32 pixels - and every activation fragment read from LDS feeds up to six MFMAs (three of each layer) instead of
//     three: this part IS the 64-cout tile body of conv_mfma.hip with [W_k ; W_{k+1}] as its two cout blocks;
//   * computes the one-pixel RING of x_k around the tile (34 + 34 + 16 + 16 = 100 pixels) as ONE more 32-pixel MFMA group per wave: a
//     lane of that group is any pixel of the ring (per-lane LDS addresses make the gather free), so the halo costs 4 pixel groups on top
//     of 16 rows x 32 columns, every wave carries the same load, and all 32 lanes of the tile's MFMAs are output pixels;
//   * activates x_k, rounds it to fp16 exactly as the store does, keeps it in LDS (positions outside the image are ZEROS: they are
//     conv_{k+1}'s padding, not conv_k evaluated on padded input), writes the tile's own 16 x 32 pixels to memory (conv5 and the later
//     layers read them) and finishes conv_{k+1} with two K-chunks from LDS.
// Per pixel and RDB the planes read drop from 4+6+8+10 = 28 to 4+8 = 12 (conv5's 12 stay), the launches from 5 to 3.
// MFMA work against four launches: conv_k x 20/16 (the ring), conv_{k+1} x 1: + 10 % on (conv1, conv2), + 11 % on (conv3, conv4).
//
// Geometry (fp16, 4 waves, TWO workgroups per CU like the kernel it replaces).  All LDS images of a tile share ONE coordinate system:
// buffer (row r, column c) <-> image (y0 - 2 + r, x0 - 2 + c), rows of 36 pixels:
//   input halo tile: rows 0..19, columns 0..35, one per K-chunk (22.5 KB, double buffered);
//   x_k:             rows 1..18, columns 1..34, plane 0 in tile buffer 0 and plane 1 in buffer 1 once the input chunks are done;
//   wave w owns output rows y0 + 4w .. 4w+3 of BOTH layers (they read the same six buffer rows 4w+1 .. 4w+6 at columns lane+1+dx)
//   plus ring group w; conv_{k+1}'s reads of x_k use the very same addresses as its reads of the input tile.
// Weights: a five-slot ring of 6 KB units in LDS.  A unit is one tap column (dx) of one K-chunk of both layers ([W_k: 3 KB][W_{k+1}:
// 3 KB], DMA'd from the two layers' own packed blobs - pack.cpp, nothing is repacked); conv_{k+1}'s two x_k chunks are three more units.
// Two 22.5 KB tile buffers + two 18 KB weight stages would be 81.3 KB - one workgroup per CU; the ring (30 KB) needs one more
// (LDS-only) barrier per K-chunk: after tap column 0 every wave is past the oldest unit and its slot takes the next chunk's last unit.
//
// Results are BIT-IDENTICAL to the four launches on the LDS-weights kernel (tests/test_gpu_dense_pair.py): same packed fragments, fp32
// accumulators start from the bias, MFMAs per output in (K-chunk, dx, dy) order, LeakyReLU as max(t, slope t), round-to-nearest fp16.
#include "common.h"
#include <cmath>
#include "conv_tile.h"

namespace ss4k {
namespace dense {

constexpr int NW = 4, MB = 4, TH = NW * MB;
constexpr int XH = TH + 4, XW = TW + 4;             // rows / columns of a tile buffer image
constexpr int REC = 32, SPR = 2;
constexpr int ROWX = XW * REC;                      // 1152 bytes per image row
constexpr int XT_SLOTS = XH * XW * SPR;             // 1440 16-byte slots
constexpr int XT_BYTES = XT_SLOTS * 16;             // 23040
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 23 wave-level DMA instructions (the last one half full)
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 6
constexpr int WU = 6 * 1024, NSLOT = 5;             // weight unit, ring slots
constexpr int NDMA = 11;                            // DMA slots per wave and chunk: 6 tile pieces, 3 for weight units 0 and 1, 2 for unit 2
constexpr int W_OFF = 2 * XT_BYTES, B_OFF = W_OFF + NSLOT * WU, R_OFF = B_OFF + 64 * 4;   // R_OFF: every thread's ring-table entry
constexpr size_t LDS_BYTES = R_OFF + 64 * NW * 4;
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert((TH + 3) * ROWX <= XT_BYTES, "x_k's plane image fits a tile buffer");

// Which pixel of the one-pixel ring of x_k around the tile a lane of the ring group computes: [wave][lane & 31] = j | i << 8 | live << 16 in
// x_k coordinates (j, i) = tile-image pixel (j + 1, i + 1); top row j = 0 and bottom row j = 17 (i = 0..33), left column i = 0 and right
// column i = 33 (j = 1..16): 100 pixels on 4 x 32 lanes, an idle lane repeats a live lane's pixel (same LDS address: a broadcast).
// Chosen for ds_read_b128's real lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) by tools/costing/ring_table.py: a column's
// pixels share two bank slots, so zero conflicts is out of reach with this image - 45 extra LDS cycles over the 9 taps x 8 groups
// against 126 for round 4's formula (two column pixels + nine row pixels per 16-lane HALF).  Placement only: any table that covers the
// ring gives the same bits.
__constant__ uint32_t RING_TAB[128] = {
// tools/costing/ring_table.py: 45 extra LDS cycles over the 9 taps x 8 lane groups (round 4's formula: 126)
  0x10000, 0x10f00, 0x11000, 0x11500, 0x10300, 0x10400, 0x10600, 0x10a00, 0x11d00, 0x11e00, 0x10005, 0x12106, 0x11600, 0x11900, 0x12100, 0x10003,
  0x12107, 0x1000a, 0x1000c, 0x10511, 0x12108, 0x10009, 0x10111, 0x10611, 0x11311, 0x11611, 0x11d11, 0x00000, 0x10911, 0x10a11, 0x10b11, 0x11111,
  0x10700, 0x10c00, 0x10d00, 0x11b00, 0x10500, 0x10900, 0x11200, 0x12104, 0x10006, 0x1000b, 0x10311, 0x00500, 0x12103, 0x1210a, 0x1210c, 0x1000f,
  0x00500, 0x00500, 0x00500, 0x00500, 0x1210f, 0x10010, 0x10411, 0x10e11, 0x10f11, 0x11011, 0x11a11, 0x11e11, 0x00500, 0x00500, 0x00500, 0x00500,
  0x10200, 0x11a00, 0x12102, 0x12105, 0x11100, 0x11800, 0x11c00, 0x11f00, 0x10002, 0x10007, 0x1210b, 0x1000d, 0x1000e, 0x10211, 0x10711, 0x10d11,
  0x11211, 0x11411, 0x11511, 0x11711, 0x11b11, 0x11f11, 0x12011, 0x00200, 0x00200, 0x00200, 0x00200, 0x00200, 0x11811, 0x12111, 0x01100, 0x01100,
  0x10800, 0x10b00, 0x11300, 0x11400, 0x10100, 0x10e00, 0x11700, 0x10008, 0x12109, 0x10011, 0x11c11, 0x00100, 0x12000, 0x10001, 0x12101, 0x10004,
  0x00100, 0x00100, 0x00100, 0x00100, 0x1210d, 0x1210e, 0x12110, 0x10811, 0x10c11, 0x11911, 0x00800, 0x00800, 0x00100, 0x00100, 0x00100, 0x00100,
};

// LeakyReLU with a slope in [0, 1] as max(t, slope t) - the value conv_mfma.hip's epilogue computes with fmaxf - of the 16 values of an
// accumulator: the products two at a time (v_pk_mul_f32), the maximum as ONE v_max_f32 each (asm: fmaxf, and v_med3(t, slope t, +inf)
// which the compiler folds back to it, cost two - in IEEE mode a possible signalling NaN is first quieted with v_max(t, t); the
// hardware instruction returns the same value for every non-NaN input).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lrelu16(const f32x16& acc, float slope, float* v) {
  const f32x2 s2 = {slope, slope};
#pragma unroll
  for (int i = 0; i < 16; i += 2) {
    const f32x2 t = {acc[i], acc[i + 1]};
    const f32x2 n = t * s2;
    asm("v_max_f32 %0, %1, %2" : "=v"(v[i]) : "v"(t.x), "v"(n.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v[i + 1]) : "v"(t.y), "v"(n.y));
  }
}

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
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  // STAMP: [0] tile setup + accumulator init, [1] x-phase reads + MFMAs + DMA issue, [2] x-phase vmcnt wait, [3] x-phase barriers, [4] x_k
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
  // byte offset of pixel (r, c)'s 16-byte half `lh` inside a tile buffer image
  auto img_off = [&](int r, int c) { return ((r * XW + c) * SPR + (lh ^ swz(c))) * 16; };
  // operand read base of both layers' 16 x 32 pixels: buffer row 4w + 1 (+ ir), column lane + 1 + dx
  int rd_base[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) rd_base[dx] = img_off(wave * MB + 1, lr + 1 + dx);
  // The ring of x_k around the tile: this lane's pixel from RING_TAB, parked in LDS (one slot per thread) and re-read where it is needed -
  // cheaper than a register through the MFMA loops.  (The slot is read by its own thread only: no barrier.)
  uint32_t* ring_slot = reinterpret_cast<uint32_t*>(smem + R_OFF) + tid;
  *ring_slot = RING_TAB[wave * 32 + lr];
  auto ring_pixel = [&]() -> int {   // j | i << 8 | live << 16
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(lds0 + R_OFF + tid * 4)) : "memory");
    return (int)v;
  };
  int rd_h[3];   // x_k (j, i) reads buffer (j + dy, i + dx)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) { const int hp = ring_pixel(); rd_h[dx] = img_off(hp & 0xff, ((hp >> 8) & 0xff) + dx); }

  // per-lane source of every halo-tile DMA instruction of this wave: byte offset inside a plane (32 bits: the host sends bigger
  // planes down the four-launch route), OOB = the zero page.  LDS slot s = 64k + lane holds pixel (row, x) = (s / 2) divmod 36, source
  // channel group (s & 1) ^ swz(x).  Recomputed per tile (six divisions by a constant) instead of kept in registers.
  uint32_t src_off[DMA_PER_WAVE];
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
      const int s = (wave + NW * j) * 64 + lane;
      const int p = s >> 1, gq = s & 1;
      const int row = p / XW, x = p - row * XW;
      const int iy = y0 - 2 + row, ix = x0 - 2 + x;
      const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      src_off[j] = ok ? ((uint32_t)(n * a.H + iy) * (uint32_t)a.W + (uint32_t)ix) * REC + (uint32_t)((gq ^ swz(x)) * 16) : OOB;
    }
  };
  auto plane_of = [&](int c) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };

  // ---- prefetch state.  A prefetch is an optional halo tile (23 instructions) and up to three weight units (6 each) of chunk `pf_cn` of
  // conv_k / conv_{k+1} (pf_cn < K1: unit g = tap column g of both layers) or of conv_{k+1}'s x_k chunks (pf_cn == K1: unit g = the g-th
  // 6 KB of its last two chunks); units go to ring slots pf_slot, pf_slot + 1, pf_slot + 2 (mod 5).
  const char* pf_plane = nullptr; uint32_t pf_tdst = 0; int pf_cn = 0, pf_slot = 0; bool pf_tile = false, pf_w01 = false, pf_w2 = false;
  auto slot_add = [](int s, int k) { const int t = s + k; return t >= NSLOT ? t - NSLOT : t; };
  auto unit_piece = [&](int u, int piece) {   // DMA of 1 KB piece `piece` (0..5) of unit u of the prefetched chunk
    const char* src = pf_cn < K1 ? (piece < 3 ? a.w1 : a.w2) + (size_t)((pf_cn * 9 + u * 3 + (piece < 3 ? piece : piece - 3)) * 1024)
                                 : a.w2 + (size_t)((K1 * 9 + u * 6 + piece) * 1024);
    dma16(src + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + W_OFF + slot_add(pf_slot, u) * WU + piece * 1024));
  };
  auto tile_piece = [&](int j) {
    const int k = wave + NW * j;
    if (k < XT_DMA) {
      const char* src = src_off[j] != OOB ? pf_plane + src_off[j] : a.zero_page + (lane & 3) * 16;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(pf_tdst + k * 1024);
      if (k * 64 + lane < XT_SLOTS) dma16(src, dst);   // lanes past the tile's last slot are masked off (EXEC)
    }
  };
  // DMA slots of a chunk's MFMA stream, idx 0..10.  The halo tile first (its pieces come from beyond L2 and take longest to land): idx
  // 0..5, all in tap column 0; then the weights (L2-resident): units 0, 1 on idx 6..8 and unit 2 on idx 9, 10 - its slot is free only after
  // the barrier that follows tap column 0, and these slots lie in column 1.  Column 2 issues nothing: its MFMAs cover the flight time.
  auto dma_op = [&](int idx) {
    if (idx < DMA_PER_WAVE) { if (pf_tile) tile_piece(idx); }
    else if (idx < DMA_PER_WAVE + 3) { const int k = wave + NW * (idx - DMA_PER_WAVE); if (pf_w01) unit_piece(k / 6, k % 6); }
    else if (idx < NDMA) { const int k = wave + NW * (idx - DMA_PER_WAVE - 3); if (pf_w2 && k < 6) unit_piece(2, k); }
  };
  // slot after MFMA pair m of tap column g: column 0 after every second pair (6 slots), column 1 after pairs 1, 3, 5, 7, 9 (5 slots)
  auto slot_of = [](int g, int m) { return (m & 1) ? (g == 0 ? m / 2 : g == 1 && m / 2 < 5 ? 6 + m / 2 : -1) : -1; };

  float* bias_lds = reinterpret_cast<float*>(smem + B_OFF);  // [conv_k 32][conv_{k+1} 32]
  if (tid < 64) bias_lds[tid] = tid < 32 ? a.bias1[tid] : a.bias2[tid - 32];

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  int slot0 = 0;                                     // ring slot of the tile's first weight unit (the tile buffer is always 0)
  pf_tile = true; pf_w01 = true; pf_w2 = true; pf_plane = plane_of(0); pf_tdst = lds0; pf_cn = 0; pf_slot = 0;
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(i);
  dma_wait();
  __syncthreads();

  const float slope = a.slope;
  const int lane16 = lane * 16;
  while (true) {
    f32x16 acc1[MB], acch, acc2[MB];
    {
      // accumulators start from the bias: 16-byte LDS reads (every lane of a half-wave reads the same four floats: a broadcast)
      // straight into the accumulator registers - 36 reads instead of 8 reads + 144 register moves
      // (as asm: left to the compiler the identical addresses are read once and copied 144 times)
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const uint32_t bl = lds0 + B_OFF + lh * 32;
      auto init = [&](f32x16& acc, int layer) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4v b;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(bl), "i"(layer * 128 + 64 * (q >> 1) + 16 * (q & 1)));
          acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
        }
      };
      init(acch, 0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) { init(acc1[mb], 0); init(acc2[mb], 1); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);
    int tb = 0, slot = slot0;
    stamp(0);

    // ------------------------------------------------ x-phase: the planes both layers read, one halo tile per K-chunk
#pragma unroll 1
    for (int c = 0; c < K1; ++c) {
      // next chunk's tile + its three weight units (the last x chunk: conv_{k+1}'s x_k-chunk weights, no tile)
      pf_tile = c + 1 < K1; pf_w01 = true; pf_w2 = true; pf_cn = c + 1; pf_slot = slot_add(slot, 3);
      pf_plane = plane_of(c + 1 < K1 ? c + 1 : 0); pf_tdst = lds0 + (tb ^ 1) * XT_BYTES;
      const char* tbp = smem + tb * XT_BYTES;
      // Fragments.  An input row's fragment is used by ONE step of the (dx, row) walk only (by the <= 3 MFMAs per layer that read row ir
      // at tap column dx), so three rotating registers hold the rows of steps t, t + 1, t + 2 and step t + 3 is loaded as soon as
      // step t's MFMAs have issued.  The ring group's nine fragments (one per tap, one every second row) share one register.  The
      // weight fragments of a tap column stay for its six steps and are refilled in place for the next column after their last MFMA.
      uint4 wf1[3], wf2[3], af[3], hf;
      auto af_load = [&](int t) { return *reinterpret_cast<const uint4*>(tbp + rd_base[t / (MB + 2)] + (t % (MB + 2)) * ROWX); };
      auto hf_load = [&](int q) { return *reinterpret_cast<const uint4*>(tbp + rd_h[q / 3] + (q % 3) * ROWX); };
      const char* wbp = smem + W_OFF + slot * WU + lane16;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        wf1[dy] = *reinterpret_cast<const uint4*>(wbp + dy * 1024);
        wf2[dy] = *reinterpret_cast<const uint4*>(wbp + 3072 + dy * 1024);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = af_load(t);
      hf = hf_load(0);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool more = g + 1 < 3;
        const char* wbn = smem + W_OFF + slot_add(slot, g + 1) * WU + lane16;   // next tap column's unit
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
              acc1[mb] = mma<__half>(wf1[dy], af[t % 3], acc1[mb]);
              acc2[mb] = mma<__half>(wf2[dy], af[t % 3], acc2[mb]);
              if (slot_of(g, m) >= 0) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(slot_of(g, m));
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
          if (ir & 1) {   // ring group: tap (dx = g, dy = ir / 2) after rows 1, 3, 5
            const int q = g * 3 + (ir >> 1);
            acch = mma<__half>(wf1[ir >> 1], hf, acch);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 9) hf = hf_load(q + 1);   // (needed two rows of MFMAs from here)
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 3 < 3 * (MB + 2)) af[t % 3] = af_load(t + 3);
          if (more) {   // weights of the next tap column, in place, as soon as a register's last MFMA of this column has issued
            if (ir >= MB - 1) wf2[ir - (MB - 1)] = *reinterpret_cast<const uint4*>(wbn + 3072 + (ir - (MB - 1)) * 1024);
            if (ir == MB - 1) wf1[0] = *reinterpret_cast<const uint4*>(wbn);            // (the ring group used dy = 0 after row 1,
            if (ir == MB) wf1[1] = *reinterpret_cast<const uint4*>(wbn + 1024);         //  dy = 1 after row 3,
            if (ir == MB + 1) wf1[2] = *reinterpret_cast<const uint4*>(wbn + 2048);     //  dy = 2 after row 5)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (g == 0) { stamp(1); lds_barrier(); stamp(3); }   // every wave is past this chunk's first unit: its slot takes the next chunk's last
      }
      stamp(1);
      dma_wait();        // the next chunk's tile / weights have landed
      stamp(2);
      __syncthreads();   // and every wave is done with this chunk's buffers
      stamp(3);
      tb ^= 1; slot = slot_add(slot, 3);
    }
    // here: tb == 0 (K1 is even); ring slots slot, slot + 1, slot + 2 hold conv_{k+1}'s chunks K1 and K1 + 1 (6 x 3 KB in order); both
    // tile buffers are free

    // ------------------------------------------------ x_k: activation, fp16, -> LDS (zeros outside the image) and -> memory (own pixels)
    {
      int lhe = lh;
      asm volatile("" : "+v"(lhe));   // re-derive the pointers per tile (hoisted they cost registers through the MFMA loops)
      char* o1 = a.out1 + (size_t)a.out1_plane0 * a.out1_plane_bytes + (size_t)lhe * 16;
      // x_k pixel at buffer (r, c) = image (cur_y0 - 2 + r, cur_x0 - 2 + c)
      auto put = [&](const f32x16& acc, int r, int c, bool own, bool live) {
        const int y = cur_y0 - 2 + r, x = cur_x0 - 2 + c;
        float v[16];
        lrelu16(acc, slope, v);
        uint4 h0, h1;
        __half* p0 = reinterpret_cast<__half*>(&h0);
        __half* p1 = reinterpret_cast<__half*>(&h1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { p0[i] = __float2half(v[i]); p1[i] = __float2half(v[8 + i]); }
        const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
        const uint32_t msk = in ? 0xFFFFFFFFu : 0u;   // outside the image: zeros (conv_{k+1}'s padding)
        char* l = smem + ((r * XW + c) * SPR + (lhe ^ swz(c))) * 16;
        if (live) {
          *reinterpret_cast<uint4*>(l) = make_uint4(h0.x & msk, h0.y & msk, h0.z & msk, h0.w & msk);
          *reinterpret_cast<uint4*>(l + XT_BYTES) = make_uint4(h1.x & msk, h1.y & msk, h1.z & msk, h1.w & msk);
        }
        if (own && in) {
          char* o = o1 + ((size_t)(cur_n * a.H + y) * a.W + x) * REC;
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&h0), reinterpret_cast<u32x4*>(o));
          __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&h1), reinterpret_cast<u32x4*>(o + a.out1_plane_bytes));
        }
      };
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) put(acc1[mb], wave * MB + 2 + mb, lr + 2, true, true);
      const int hpix = ring_pixel();
      put(acch, (hpix & 0xff) + 1, ((hpix >> 8) & 0xff) + 1, false, (hpix >> 16) != 0);
    }
    stamp(4);
    lds_barrier();   // x_k is visible; (its stores to memory stay in flight)
    stamp(5);

    // ------------------------------------------------ conv_{k+1}'s last two K-chunks: x_k from LDS, at the input tile's addresses
    if (next_tile >= 0) setup_tile(next_tile, n, y0, x0);
    {
      uint4 wf[3], af[3];
      // tap column g of chunk K1 + h: the (3h + g)-th 3 KB of the three units
      auto wptr = [&](int h, int g) { const int e = 3 * h + g; return smem + W_OFF + slot_add(slot, e >> 1) * WU + (e & 1) * 3072 + lane16; };
      auto af_load = [&](int h, int t) { return *reinterpret_cast<const uint4*>(smem + h * XT_BYTES + rd_base[t / (MB + 2)] + (t % (MB + 2)) * ROWX); };
      auto first_frags = [&](int h) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) wf[dy] = *reinterpret_cast<const uint4*>(wptr(h, 0) + dy * 1024);
#pragma unroll
        for (int t = 0; t < 3; ++t) af[t] = af_load(h, t);
      };
      first_frags(0);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // chunk K1: the next tile's first two weight units go into the two free slots; chunk K1 + 1 (every wave is past the ring's
        // oldest unit and past plane 0's image): its third unit and its first halo tile (buffer 0)
        pf_tile = h == 1 && next_tile >= 0; pf_w01 = h == 0 && next_tile >= 0; pf_w2 = pf_tile;
        pf_plane = plane_of(0); pf_tdst = lds0; pf_cn = 0; pf_slot = slot_add(slot, 3);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const bool more = g + 1 < 3;
          const char* wbn = wptr(h, more ? g + 1 : g);
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
                if (slot_of(g, m) >= 0) {
                  __builtin_amdgcn_sched_barrier(0);
                  dma_op(slot_of(g, m));
                  __builtin_amdgcn_sched_barrier(0);
                }
                ++m;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < 3 * (MB + 2)) af[t % 3] = af_load(h, t + 3);
            if (more && ir >= MB - 1) wf[ir - (MB - 1)] = *reinterpret_cast<const uint4*>(wbn + (ir - (MB - 1)) * 1024);
            __builtin_amdgcn_sched_barrier(0);
          }
          __builtin_amdgcn_s_setprio(0);
        }
        stamp(6);
        if (h == 0) {
          first_frags(1);   // plane 1's image and its weights are visible since the barrier before chunk K1: read ahead of the barrier
          lds_barrier(); stamp(7);
        }
      }
    }
    // hand the buffers to the next tile BEFORE the epilogue (conv_mfma.hip): the stores drain under its first chunk's MFMAs
    if (next_tile >= 0) {
      dma_wait();
      stamp(8);
      __syncthreads();
      stamp(9);
      slot0 = slot_add(slot, 3);
    }

    // ------------------------------------------------ conv_{k+1}'s epilogue: activation, fp16, -> memory
    {
      const int x = cur_x0 + lr;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));
      char* o2 = a.out2 + (size_t)a.out2_plane0 * a.out2_plane_bytes + (size_t)lhe * 16;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int y = cur_y0 + wave * MB + mb;
        float v[16];
        lrelu16(acc2[mb], slope, v);
        if (x < a.W && y < a.H) {
          char* o = o2 + ((size_t)(cur_n * a.H + y) * a.W + x) * REC;
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

// =====================================================================================================================
// conv3x3_wide_kernel: ONE 3x3 layer with 64 output channels per workgroup on the fused kernel's machinery (round 4) - the x-phase
// above with the two cout blocks of one layer as its accumulator sets.  Against conv_mfma.hip's <__half,2,4,4> build (same tile,
// same MFMA order per output: bit-identical results): three rotating activation-fragment registers instead of a fragment per row, the
// halo tile's DMAs issued first and none in the last tap column, accumulators initialised by LDS reads, packed epilogue arithmetic,
// weights as six 6 KB units (= two double-buffered stages addressed by tap column).  18 x 34 halo tile: 76 KB, two workgroups per CU.
// Plain layout epilogue (activation, alpha, up to two residuals, in-place safe), nearest-x2 upsampled input as an address mode.
namespace wide {

constexpr int NW = 4, MB = 4, TH = NW * MB;
constexpr int XH = TH + 2, XW = TW + 2;
constexpr int REC = 32, SPR = 2;
constexpr int ROWX = XW * REC;                      // 1088
constexpr int XT_SLOTS = XH * XW * SPR;             // 1224
constexpr int XT_BYTES = XT_SLOTS * 16;             // 19584
constexpr int XT_DMA = (XT_SLOTS + 63) / 64;        // 20
constexpr int DMA_PER_WAVE = (XT_DMA + NW - 1) / NW;  // 5
constexpr int WU = 6 * 1024, NSLOT = 6;             // a unit = one tap column of one chunk: [dy][cout block][lane] fragments, as packed
constexpr int NDMA = DMA_PER_WAVE + 5;
constexpr int W_OFF = 2 * XT_BYTES, B_OFF = W_OFF + NSLOT * WU;
constexpr size_t LDS_BYTES = B_OFF + 2 * 64 * 4;    // + bias and slope of the group's 64 couts
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");

// UPS: the input is the nearest-x2 up-sampling of a half-resolution tensor (RRDBNet's conv_up1 / conv_up2) and the layer asked for the
// pre-summed form.  Two of the three input rows an output row reads are then the SAME low-resolution row (rows y, y + 1 for an even y;
// y - 1, y for an odd one): their two MFMAs per tap column become one with the two weight fragments added (fp16, four v_pk_add_f16 per
// fragment and tap column) - 6 instead of 9 MFMAs per output pixel and K-chunk.  Not bit-identical to the direct form (one more fp16
// rounding of a weight sum, a different order of fp32 additions); SS4K_MODEL_NO_UPS_PRESUM selects the direct form.
// RL: res1 is the layer's own input tensor and the first four K-chunks are its planes (conv5 of an RDB: out = conv * alpha + x, no
// activation): x's centre pixels are in LDS as part of those chunks, so they are added to the accumulators there - one more MFMA per
// output row and chunk with a (1 / alpha) * I fragment on the centre tap, as conv_rs.hip does - instead of being read from memory a
// second time in the epilogue (measured on the layer in isolation: 225 -> 210 us per 4 frames, tools/conv5_routes.py).
template <bool UPS, bool RL = false>
__global__ __launch_bounds__(64 * NW, 2) void conv3x3_wide_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = blockIdx.y;
  const int K = a.nchunks0 + a.nchunks1;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const char* wbase = reinterpret_cast<const char*>(a.wpk) + (size_t)grp * K * (3 * WU);
  const int Hs = a.ups2 ? (a.H >> 1) : a.H, Ws = a.ups2 ? (a.W >> 1) : a.W;

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
  auto swz = [](int x) { return (x >> 3) & 1; };
  int rd_base[3];   // buffer (row r, column c) <-> image (y0 - 1 + r, x0 - 1 + c); wave w reads rows 4w .. 4w+5 at columns lane + dx
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) { const int c = lr + dx; rd_base[dx] = (((wave * MB) * XW + c) * SPR + (lh ^ swz(c))) * 16; }

  uint32_t src_off[DMA_PER_WAVE];
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
      const int s = (wave + NW * j) * 64 + lane;
      const int p = s >> 1, gq = s & 1;
      const int row = p / XW, x = p - row * XW;
      const int iy = y0 - 1 + row, ix = x0 - 1 + x;
      const bool ok = s < XT_SLOTS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int sy = a.ups2 ? (iy >> 1) : iy, sx = a.ups2 ? (ix >> 1) : ix;
      src_off[j] = ok ? ((uint32_t)(n * Hs + sy) * (uint32_t)Ws + (uint32_t)sx) * REC + (uint32_t)((gq ^ swz(x)) * 16) : OOB;
    }
  };
  auto plane_of = [&](int c) {
    return (c < a.nchunks0) ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                            : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  };
  const char* pf_plane = nullptr; const char* pf_w = nullptr; uint32_t pf_tdst = 0; int pf_slot = 0; bool pf_on = false;
  auto slot_add = [](int s, int k) { const int t = s + k; return t >= NSLOT ? t - NSLOT : t; };
  // DMA slots idx 0..9: the halo tile first (5), then the chunk's 18 KB of weights (18 pieces over 5 slots)
  auto dma_op = [&](int idx) {
    if (!pf_on) return;
    if (idx < DMA_PER_WAVE) {
      const int k = wave + NW * idx;
      if (k < XT_DMA) {
        const char* src = src_off[idx] != OOB ? pf_plane + src_off[idx] : a.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(pf_tdst + k * 1024);
        if (k * 64 + lane < XT_SLOTS) dma16(src, dst);
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - DMA_PER_WAVE);   // piece 0..17 of the chunk = unit k / 6, piece k % 6
      if (k < 18) dma16(pf_w + k * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds0 + W_OFF + slot_add(pf_slot, k / 6) * WU + (k % 6) * 1024));
    }
  };
  // DMA slot after MFMA pair m of tap column g (12 pairs per column; UPS: 8, and a slot after every pair)
  auto slot_of = [](int g, int m) {
    if (UPS) return g == 0 ? m : g == 1 ? 8 + m : -1;
    return (m & 1) ? (g == 0 ? m / 2 : g == 1 && m / 2 < 5 ? 6 + m / 2 : -1) : -1;
  };

  float* epi_lds = reinterpret_cast<float*>(smem + B_OFF);   // [64 bias][64 slope]
  if (tid < 64) {
    const int v = grp * 64 + tid;
    epi_lds[tid] = v < a.cout_pad ? a.bias[v] : 0.f;
    epi_lds[64 + tid] = a.act == ACT_PRELU ? (v < a.cout_pad ? a.prelu[v] : 1.f) : (a.act == ACT_LRELU ? a.slope : (a.act == ACT_RELU6 ? 0.f : 1.f));
  }

  int kt = 0;
  int tile = tile_of(0);
  if (tile < 0) return;
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  int tb = 0, slot = 0;
  pf_on = true; pf_plane = plane_of(0); pf_tdst = lds0; pf_w = wbase; pf_slot = 0;
#pragma unroll
  for (int i = 0; i < 12; ++i) if (i < NDMA) dma_op(i);
  dma_wait();
  __syncthreads();
  const int lane16 = lane * 16;
  // RL: A fragments of (1 / alpha) * I for the two planes of a 32-cout block.  MFMA row rho of a block is virtual cout
  // v(rho) = 16 (rho >> 4) + 8 ((rho >> 2) & 1) + (rho & 3) + 4 ((rho >> 3) & 1) (pack.cpp); this lane holds k = 8 lh .. 8 lh + 7
  uint4 iid[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
  if constexpr (RL) {
    const int rho = lane & 31, v = 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + (rho & 3) + 4 * ((rho >> 3) & 1);
    const unsigned hv = (unsigned)__half_as_ushort(__float2half(1.f / a.alpha));
    auto frag = [&](int p) {
      const int e = v - 16 * p - 8 * lh;   // element of this lane's 8 k-values that meets row rho on plane p, if any
      const uint32_t w = hv << (16 * (e & 1));
      return make_uint4((e >> 1) == 0 ? w : 0u, (e >> 1) == 1 ? w : 0u, (e >> 1) == 2 ? w : 0u, (e >> 1) == 3 ? w : 0u);
    };
    iid[0] = frag(0); iid[1] = frag(1);
  }

  while (true) {
    f32x16 acc[2][MB];
    {
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const uint32_t bl = lds0 + B_OFF + lh * 32;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4v b;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b) : "v"(bl), "i"(nb * 128 + 64 * (q >> 1) + 16 * (q & 1)));
            acc[nb][mb][4 * q] = b.x; acc[nb][mb][4 * q + 1] = b.y; acc[nb][mb][4 * q + 2] = b.z; acc[nb][mb][4 * q + 3] = b.w;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0;
    const int next_tile = tile_of(kt + 1);

#pragma unroll 1
    for (int c = 0; c < K; ++c) {
      if (c + 1 < K) { pf_on = true; pf_plane = plane_of(c + 1); pf_w = wbase + (size_t)(c + 1) * (3 * WU); }
      else if (next_tile >= 0) { setup_tile(next_tile, n, y0, x0); pf_on = true; pf_plane = plane_of(0); pf_w = wbase; }
      else pf_on = false;
      pf_tdst = lds0 + (tb ^ 1) * XT_BYTES; pf_slot = slot_add(slot, 3);
      const char* tbp = smem + tb * XT_BYTES;
      uint4 wf[3][2], af[3];
      auto af_load = [&](int t) { return *reinterpret_cast<const uint4*>(tbp + rd_base[t / (MB + 2)] + (t % (MB + 2)) * ROWX); };
      const char* wbp = smem + W_OFF + slot * WU + lane16;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[dy][nb] = *reinterpret_cast<const uint4*>(wbp + (dy * 2 + nb) * 1024);
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = af_load(t);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const bool more = g + 1 < 3;
        const char* wbn = smem + W_OFF + slot_add(slot, g + 1) * WU + lane16;
        int m = 0;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // UPS: the summed fragments of this tap column (this wave's rows start at an even output row: tile rows and 4w are even)
        uint4 ws01[2], ws12[2];
        if constexpr (UPS) {
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          auto addh = [](const uint4& p, const uint4& q) {
            uint4 r;
            const uint32_t* pp = &p.x; const uint32_t* qq = &q.x; uint32_t* rr = &r.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, pp[k]) + __builtin_bit_cast(h2, qq[k]));
            return r;
          };
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) { ws01[nb] = addh(wf[0][nb], wf[1][nb]); ws12[nb] = addh(wf[1][nb], wf[2][nb]); }
        }
#pragma unroll
        for (int ir = 0; ir < MB + 2; ++ir) {
          const int t = g * (MB + 2) + ir;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int mb = ir - dy;
            // UPS, even output row: rows (y - 1) and (y, y + 1 summed): dy 0 plain, dy 1 with w1 + w2, dy 2 skipped;
            //      odd output row:  (y - 1, y summed) and y + 1:        dy 0 with w0 + w1, dy 1 skipped, dy 2 plain
            const bool skip = UPS && ((mb & 1) ? dy == 1 : dy == 2);
            if (mb >= 0 && mb < MB && !skip) {
              const uint4& w0 = !UPS ? wf[dy][0] : (mb & 1) ? (dy == 0 ? ws01[0] : wf[2][0]) : (dy == 0 ? wf[0][0] : ws12[0]);
              const uint4& w1 = !UPS ? wf[dy][1] : (mb & 1) ? (dy == 0 ? ws01[1] : wf[2][1]) : (dy == 0 ? wf[0][1] : ws12[1]);
              acc[0][mb] = mma<__half>(w0, af[t % 3], acc[0][mb]);
              acc[1][mb] = mma<__half>(w1, af[t % 3], acc[1][mb]);
              if constexpr (RL) {
                if (g == 1 && dy == 1 && c < 4) {   // centre tap of an x chunk: + x / alpha on the block that holds this plane's couts
                  const uint32_t pm = (c & 1) ? 0xFFFFFFFFu : 0u;   // plane 1 or plane 0 of the block (wave-uniform)
                  const uint4 idf = make_uint4((iid[1].x & pm) | (iid[0].x & ~pm), (iid[1].y & pm) | (iid[0].y & ~pm),
                                               (iid[1].z & pm) | (iid[0].z & ~pm), (iid[1].w & pm) | (iid[0].w & ~pm));
                  if (c < 2) acc[0][mb] = mma<__half>(idf, af[t % 3], acc[0][mb]);
                  else acc[1][mb] = mma<__half>(idf, af[t % 3], acc[1][mb]);
                }
              }
              if (slot_of(g, m) >= 0 && slot_of(g, m) < NDMA) {
                __builtin_amdgcn_sched_barrier(0);
                dma_op(slot_of(g, m));
                __builtin_amdgcn_sched_barrier(0);
              }
              ++m;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + 3 < 3 * (MB + 2)) af[t % 3] = af_load(t + 3);
          if (more && ir >= MB - 1) {
            const int dy = ir - (MB - 1);
            wf[dy][0] = *reinterpret_cast<const uint4*>(wbn + (dy * 2) * 1024);
            wf[dy][1] = *reinterpret_cast<const uint4*>(wbn + (dy * 2 + 1) * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      // the next chunk (or the next tile's first chunk: the hand-over happens BEFORE the epilogue, its stores drain under MFMAs)
      if (pf_on) { dma_wait(); __syncthreads(); }
      tb ^= 1; slot = slot_add(slot, 3);
    }

    // ---------------- epilogue (conv_mfma.hip's plain-layout fast path, the same expressions: results are bit-identical) ----------------
    {
      const float alpha = a.alpha, gamma = a.gamma;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));
      const int xo = cur_x0 + lr;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int vblock = (grp * 2 + nb) * 32;
        if (vblock >= a.cout_pad) continue;
        float slope_v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + 64 + nb * 32 + 16 * (q >> 1) + 8 * lhe + 4 * (q & 1));
          slope_v[4 * q] = s4.x; slope_v[4 * q + 1] = s4.y; slope_v[4 * q + 2] = s4.z; slope_v[4 * q + 3] = s4.w;
        }
        const int opl = vblock / CW;
        const size_t sub = (size_t)lhe * 16;
        const char* r1p = (a.res1 && !RL) ? a.res1 + (size_t)(a.r1_plane0 + opl) * a.r1_plane_bytes + sub : nullptr;
        const char* r2p = a.res2 ? a.res2 + (size_t)(a.r2_plane0 + opl) * a.r2_plane_bytes + sub : nullptr;
        char* outp = a.out + (size_t)(a.out_plane0 + opl) * a.out_plane_bytes + sub;
        const size_t pix0 = ((size_t)cur_n * a.H + cur_y0 + wave * MB) * a.W + xo;
        if (!r1p && !r2p) {
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
              for (int i = 0; i < 16; ++i) { const float t = acc[nb][mb][i], neg = t * slope_v[i]; v[i] = (t >= 0.f ? t : neg) * alpha; }
            } else {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const float t = acc[nb][mb][i], st = t * slope_v[i];
                float mx;
                asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(t), "v"(st));   // = fmaxf(t, st) for every non-NaN input, one instruction
                v[i] = mx * alpha;
              }
            }
            if (ok) {
              char* o = outp + (pix0 + (size_t)mb * a.W) * REC;
              store8<__half>(o, v);
              store8<__half>(o + a.out_plane_bytes, v + 8);
            }
          }
        } else {
          constexpr int RB = 1;   // 128 accumulator registers leave room for one row of residuals at a time
#pragma unroll
          for (int mb0 = 0; mb0 < MB; mb0 += RB) {
            uint4 r1v[2], r2v[2];
            const int mb = mb0;
            const bool ok = (cur_y0 + wave * MB + mb) < a.H && xo < a.W;
            const size_t rec = (pix0 + (size_t)mb * a.W) * REC;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              r1v[hq] = (r1p && ok) ? *reinterpret_cast<const uint4*>(r1p + hq * (size_t)a.r1_plane_bytes + rec) : make_uint4(0, 0, 0, 0);
              r2v[hq] = (r2p && ok) ? *reinterpret_cast<const uint4*>(r2p + hq * (size_t)a.r2_plane_bytes + rec) : make_uint4(0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float v[16], r1[16], r2[16];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              load8<__half>(reinterpret_cast<const char*>(&r1v[hq]), r1 + 8 * hq);
              load8<__half>(reinterpret_cast<const char*>(&r2v[hq]), r2 + 8 * hq);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              float t = acc[nb][mb][i];
              const float neg = t * slope_v[i];
              t = t >= 0.f ? t : neg;
              if (a.act == ACT_RELU6) t = fminf(t, 6.f);
              t = t * alpha + r1[i];
              v[i] = t * gamma + r2[i];
            }
            if (ok) {
              char* o = outp + rec;
              store8<__half>(o, v);
              store8<__half>(o + a.out_plane_bytes, v + 8);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (next_tile < 0) break;
    tile = next_tile; ++kt;
  }
}

}  // namespace wide

bool conv3x3_wide_eligible(const ConvArgs& a, int dtype) {
  return dtype == SS4K_F16 && a.epi == EPI_NHWC && !a.bsvd_resid && !a.dbg && a.cout_pad >= 64 && a.cout_pad % 64 == 0 &&
         conv_plane_span_f16(a) < 4294967296.0 && (!a.ups2 || (a.H % 2 == 0 && a.W % 2 == 0));
}

void launch_conv3x3_wide(ss4k_ctx* ctx, const ConvArgs& a0, hipStream_t st) {
  using namespace wide;
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int groups = a.cout_pad / 64;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 / groups * (a.grid_share > 0.f ? a.grid_share : 1.f))));
  auto go = [&](auto kern) {
    const void* fn = reinterpret_cast<const void*>(kern);
    if (ctx->lds_attr_set.insert(fn).second)
      SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(gx, groups), dim3(64 * NW), LDS_BYTES, st, a);
  };
  // conv5 of an RDB with its residual through the matrix core (RL): res1 must be the conv's own input tensor = its first four planes
  // ... and 1 / alpha must be an fp16 number (the identity fragment carries it: RRDBNet's 0.2 -> 5.0); any other alpha would scale the skip
  // tensor by 1 +- 2^-11 on top of the output rounding, so it takes the epilogue that reads the residual from memory
  const bool alpha_exact = a.alpha != 0.f && std::fabs(__half2float(__float2half(1.f / a.alpha)) * a.alpha - 1.f) <= 1.2e-7f;
  const bool rl = a.wide_rl && a.res1 && a.act == ACT_NONE && alpha_exact && a.nchunks0 == 4 && a.cout_pad == 64 && !a.ups2 &&
                  a.res1 + (size_t)a.r1_plane0 * a.r1_plane_bytes == a.in0 + (size_t)a.in0_plane0 * a.in0_plane_bytes &&
                  a.r1_plane_bytes == a.in0_plane_bytes;
  ctx->prof_family = rl ? "wide::conv3x3_wide_kernel<RL> (64-cout tile, 32x32x16 MFMA, residual through the matrix core)"
                     : (a.ups2 && a.ups_presum) ? "wide::conv3x3_wide_kernel<UPS> (64-cout tile, 32x32x16 MFMA, nearest-x2 input, pre-summed row taps)"
                                                : "wide::conv3x3_wide_kernel (64-cout tile, 32x32x16 MFMA)";
  if (rl) go(&conv3x3_wide_kernel<false, true>);
  else if (a.ups2 && a.ups_presum) go(&conv3x3_wide_kernel<true>);
  else go(&conv3x3_wide_kernel<false>);
  SS4K_HIP(hipGetLastError());
}

bool conv3x3_dense2_eligible(int nchunks_a, int cout_pad_a, int nchunks_b, int cout_pad_b) {
  return cout_pad_a == 32 && cout_pad_b == 32 && nchunks_b == nchunks_a + 2 && nchunks_a >= 2 && nchunks_a % 2 == 0;
}

void launch_conv3x3_dense2(ss4k_ctx* ctx, const DenseArgs& a0, hipStream_t st) {
  using namespace dense;
  DenseArgs a = a0;
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "dense pair: empty grid");
  SS4K_REQUIRE((a.nchunks0 + a.nchunks1) % 2 == 0 && a.nchunks0 + a.nchunks1 >= 2, "dense pair: conv_k needs an even number of K-chunks");
  SS4K_REQUIRE(a.slope >= 0.f && a.slope <= 1.f, "dense pair: LeakyReLU slope must be in [0,1]");
  SS4K_REQUIRE((double)(a.n0 + a.N) * a.H * a.W * 32.0 < 4294967296.0, "dense pair: a plane holds at most 4 GB");
  a.tiles_x = (a.W + TW - 1) / TW; a.tiles_y = (a.H + TH - 1) / TH;
  a.zero_page = ctx->zero_page();
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu * 2 * (a.grid_share > 0.f ? a.grid_share : 1.f))));
#ifdef SS4K_DEV
  if (const char* e = std::getenv("SS4K_DENSE_GRID")) gx = std::min(ntiles, std::max(1, std::atoi(e)));   // e.g. 256: one workgroup per CU (stamps without a partner)
#endif
  ProfScope prof(ctx, st, PROF_CONV);
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
    prof.done(a0.flops);
    return;
  }
#endif
  ctx->prof_family = a.nchunks0 + a.nchunks1 == 4 ? "dense::conv3x3_dense2_kernel<4> (fused pair conv1 + conv2 of an RDB, 32x32x16 MFMA)"
                     : a.nchunks0 + a.nchunks1 == 8 ? "dense::conv3x3_dense2_kernel<8> (fused pair conv3 + conv4 of an RDB, 32x32x16 MFMA)"
                                                    : "dense::conv3x3_dense2_kernel<0> (fused layer pair, 32x32x16 MFMA)";
  switch (a.nchunks0 + a.nchunks1) {
    case 4: go(&conv3x3_dense2_kernel<4>); break;
    case 8: go(&conv3x3_dense2_kernel<8>); break;
    default: go(&conv3x3_dense2_kernel<0>); break;
  }
  SS4K_HIP(hipGetLastError());
  prof.done(a0.flops);
}

}  // namespace ss4k
