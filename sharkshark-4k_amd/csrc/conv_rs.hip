// 3x3 convolution (pad 1, stride 1), fp16 storage / fp32 accumulate, with REGISTER-STATIONARY WEIGHTS
// on v_mfma_f32_16x16x32_f16 - the production kernel of the RRDBNet / SRVGG body layers (round 2).
//
// Why this shape (measurements: tools/micro/mfma_shapes.hip, tools/micro/rs_skeleton.hip, profiles/NOTES_r01_r03.md 4.1b):
//   * on random data the chip holds a ~13 % higher clock on the 16x16x32 MFMA than on 32x32x16 at
//     equal cycles per FLOP (the conv kernel is power-bound, not issue-bound);
//   * K = 32 of that MFMA is one tap x 32 input channels = TWO 16-channel planes, so an LDS pipeline
//     stage is a 64-byte-per-pixel halo tile (38 KB for 16x32 pixels); together with double-buffered
//     weights that no longer fits two workgroups per CU - so the weights leave LDS altogether:
//   * ONE 4-wave workgroup per CU (one wave per SIMD, the whole 512-entry register file per lane).  Every
//     wave loads ITS slice of the layer's weights (all input channels x 9 taps x 16 or 32 output
//     channels, MFMA A-fragment order, 144-288 registers) once, at the start of the persistent workgroup,
//     and keeps it - mostly in the accumulation half of the register file (AGPRs), read by the MFMAs
//     directly as their A operand - for every tile it processes: no weight DMA, no weight LDS reads,
//     1/3 to 1/2 fewer L2->LDS bytes per FLOP than the LDS-weights kernel (conv_mfma.hip);
//   * LDS holds only a ring of three halo stages filled by LDS-DMA two K-chunks ahead (across tile
//     boundaries), counted s_waitcnt vmcnt(N) + one raw barrier per chunk; a tile's output stores drain
//     under the next tile's MFMAs (they are accounted for in the counted wait).
//
// Work split of the 16x32-pixel tile over the four waves, by layer shape <NCH, ROWS, CB>
// (NCH = 32-channel K-chunks, ROWS = output rows per wave, CB = 16-cout blocks per wave):
//     32 couts, cin  64/ 96/128 (RDB conv1-3) : <2|3|4, 4, 2>  4 row groups, every wave all 32 couts
//     32 couts, cin 160         (RDB conv4)   : <5, 8, 1>      2 row groups x 2 cout halves
//     64 couts, cin  64 (trunk / tail / SRVGG): <2, 8, 2>      2 row groups x 2 cout halves
//     64 couts, cin 192         (RDB conv5)   : <6, 16, 1>     every wave the whole tile, 16 couts each
// (Eight-wave variants <NCH, 4, 1, CG = 2> of the 32-cout shapes - two waves per SIMD, 256 registers each - were
// built and measured: 0.95-0.98x the LDS-weights kernel for conv1/conv2, register spills for conv3; profiles/NOTES_r01_r03.md 4.1b.)
// Data layout in HBM is conv_mfma.hip's ("planes" of 16 channels, 32-byte records), so the two kernels
// are interchangeable per layer; the fp32 parity path, the first / last layers and the BSVD epilogues stay
// on conv_mfma.hip.
#include "common.h"
#include <type_traits>

#ifndef SS4K_RS_NBF
#define SS4K_RS_NBF 3
#endif

namespace ss4k {
namespace rs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TH = 16, TW = 32, IN_H = TH + 2, IN_W = TW + 2;
constexpr int PIXB = 64;                     // LDS bytes per halo pixel: 2 planes x 16 channels x fp16
constexpr int ROWB = IN_W * PIXB;            // 2176
constexpr int TILE_SLOTS = IN_H * IN_W * 4;  // 16-byte slots of a 32-channel halo stage (2448 = 38.25 KB)
constexpr int STAGE_DMA = 40;                // wave-level 1 KB LDS-DMA instructions per stage (>= 2448 / 64 = 38.25)
constexpr int STAGE = STAGE_DMA * 1024;      // stage padded so that 4 or 8 waves issue the same number each (exact
                                             // counted waits, no EXEC masks)
constexpr int NSTAGE = 4;                    // ring: one being read, three in flight (all 160 KB of the CU)
constexpr int MAX_PLANES = 12;

// kernel arguments, reduced to what the kernel reads (scalar registers are scarce next to 100 address
// and loop scalars): plane pointers resolved on the host, outputs / residuals pre-offset to their first plane
struct RsArgs {
  const char* plane[MAX_PLANES];   // source plane of every 16-channel K-slice (two per chunk); nullptr = zeros
  const char* zero_page;
  const void* wrs; const float* bias; const float* prelu;
  char* out; size_t out_plane_bytes;
  const char* res1; size_t r1_plane_bytes;
  const char* res2; size_t r2_plane_bytes;
  int N, n0, H, W, ups2, tiles_x, tiles_y, reverse, no_band;   // n0: first frame of this launch (frame lanes)
  float slope, alpha, gamma;       // slope: LeakyReLU slope, 1 = no activation (PReLU: per channel, `prelu`)
};

__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_addr_wave_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr_wave_uniform)
               : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// acc += W * B on the matrix core.  Written as asm so that the register file of the stationary weights is OURS
// to choose: WA = the fragment lives in an AGPR quad and is read by the MFMA directly (hipcc, left alone, parks
// weights beyond 256 VGPRs in AGPRs too but copies every fragment back with four v_accvgpr_read before each
// use - two VALU issues per MFMA, which is all the issue room a 16-cycle MFMA leaves).
// Hazards (cdna_hip_programming.md 5.7): accumulate chains need no wait states; the operands come from
// ds_read / global_load (hipcc inserts those waits before the statement); the first MFMA of a tile and the
// epilogue's first read of an accumulator are fenced with s_nop below.
__device__ __forceinline__ void mfma16(f32x4& acc, const u32x4& w, const u32x4& b, bool WA) {  // WA folds after unrolling
  if (WA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(b));
}

// first MFMA of an accumulator in a tile: C = the bias quad (D is written whole, so acc needs no initialisation)
__device__ __forceinline__ void mfma16_init(f32x4& acc, const u32x4& w, const u32x4& b, const f32x4& c, bool WA) {
  if (WA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(w), "v"(b), "v"(c));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(w), "v"(b), "v"(c));
}

// 16-byte slots of a pixel are XOR-swizzled by its column so that every ds_read_b128 of a B fragment
// (16 pixels x 4 k-groups per wave) is bank-conflict free for all three tap columns
__device__ __forceinline__ int swz(int col) { return ((col >> 2) & 1) << 1; }

// PR: per-channel PReLU slopes (SRVGG); otherwise one LeakyReLU slope / identity for the whole layer.
// RL: res1 is the layer's own input tensor (conv5 of an RDB: out = conv * 0.2 + x) and there is no activation:
//     x's centre pixels are already in LDS as part of K-chunks 0 / 1, so they are added to the accumulators
//     there (scaled by 1 / alpha) instead of being read from HBM a second time in the epilogue - at one wave
//     per SIMD nothing would hide that latency.
// RES: the epilogue reads residual(s) from memory (res1 unless RL, res2); built only where a network needs it.
template <int NCH, int ROWS, int CB, int CG, bool PR, bool RL, bool RES>
__global__ __launch_bounds__(64 * (TH / ROWS) * CG, (TH / ROWS) * CG / 4) void conv3x3_rs_kernel(const RsArgs a) {
  constexpr int RG = TH / ROWS, NW = RG * CG, COUT_WG = CG * CB * 16;
  constexpr int NDMA = STAGE_DMA / NW;        // DMA instructions per wave and stage (10 with four waves, 5 with eight)
  static_assert(NW == 4 || NW == 8, "four waves (one per SIMD) or eight (two per SIMD)");
  constexpr int NV = CB * 4;                  // output channels per lane and pixel
  constexpr int NSTEP = 3 * (ROWS + 2);       // (tap column, input row) steps per K-chunk
  // K-chunks whose weight fragments live in AGPRs (<= 216 of the 256); the rest stay in VGPRs next to the
  // accumulators, B fragments and addresses
  constexpr int AGPR_W = NW == 4 ? 216 : 144;   // two waves per SIMD share the 512-entry file: 256 each
  constexpr int NA = (AGPR_W / (36 * CB)) < NCH ? (AGPR_W / (36 * CB)) : NCH;
  static_assert(NCH >= 2 && 2 * NCH <= MAX_PLANES, "the two-chunks-ahead prefetch needs at least two K-chunks per tile");
  static_assert(ROWS * 2 * CB * 4 + (NCH - NA) * 36 * CB <= 160, "VGPR budget: accumulators + VGPR-resident weights");
  static_assert(!RL || (CB == 1 && ROWS == 16), "RL is built for the conv5 shape: a wave owns one output plane of the whole tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;   // (kernel is launched with 64 * NW threads)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave % RG, cg = wave / RG;   // this wave's row group and cout group
  const int p = lane & 15, q = lane >> 4;
  const int grp = blockIdx.y;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;

  // XCD-aware persistent tile walk (same as conv_mfma.hip): workgroups b and b+8 share an XCD; each
  // XCD gets a contiguous band of tiles.  Placement only changes speed, never results.
  const bool banded = (gridDim.x % 8 == 0) && ntiles >= (int)gridDim.x && !a.no_band;
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
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) __attribute__((always_inline)) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y;
    n = a.n0 + tyn / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
  };

  // ---- this wave's weight slice -> registers (A fragments: lane (m, kq) holds cout row m, k = 8kq..8kq+7)
  u32x4 W[NCH][9][CB];
  {
    const char* wsrc = reinterpret_cast<const char*>(a.wrs) + ((size_t)(grp * CG + cg) * NCH * 9 * CB) * 1024 + lane * 16;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) W[c][t][cb] = *reinterpret_cast<const u32x4*>(wsrc + ((c * 9 + t) * CB + cb) * 1024);
    // pin the AGPR-resident fragments to their register quads now (the loads then target AGPRs directly):
    // left to the first MFMA that reads each, 216 registers of loads would be live in VGPRs at once
#pragma unroll
    for (int c = 0; c < NA; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) asm volatile("" : "+a"(W[c][t][cb]));
  }
  // epilogue constants of this lane's NV channels: cout0 + NV*q + j
  const int cout0 = grp * COUT_WG + cg * CB * 16;
  float bias_v[NV], slope_v[PR ? NV : 1];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = cout0 + NV * q + j;
    bias_v[j] = a.bias[v];
    if constexpr (PR) slope_v[j] = a.prelu[v];
  }
  if constexpr (!PR) slope_v[0] = a.slope;

  // per-lane B-fragment read bases: pixel column 16*pb + p + dx, k-group q (swizzled), wave's first row
  int rdb[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int col = 16 * pb + p + dx;
      rdb[dx][pb] = (rg * ROWS) * ROWB + col * PIXB + ((q ^ swz(col)) << 4);
    }

  // DMA plan.  LDS slot s = 64k + lane of a stage is pixel (row, col) of the halo tile, position qq; it receives
  // data group dg = qq ^ swz(col)  (dg >> 1: which plane of the pair, dg & 1: which 8-channel half).
  // Tile-independent part per DMA instruction j of this wave: row | col << 8 | dg << 16 (padding slots: row 255).
  int plan[NDMA];
#pragma unroll
  for (int j = 0; j < NDMA; ++j) {
    const int s = (wave + NW * j) * 64 + lane;
    const int pix = s >> 2, qq = s & 3;
    const int row = pix / IN_W, col = pix - row * IN_W;
    plan[j] = (s < TILE_SLOTS) ? (row | (col << 8) | ((qq ^ swz(col)) << 16)) : 0xff;
  }
  constexpr uint32_t OOB = 0xFFFFFFFFu;
  // per tile: source pixel index (inside a plane) of every slot this lane moves, OOB = zero padding
  auto tile_offsets = [&](int tn, int ty0, int tx0, uint32_t (&off)[NDMA]) {
    const int Hs = a.ups2 ? (a.H >> 1) : a.H, Ws = a.ups2 ? (a.W >> 1) : a.W;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      const int row = plan[j] & 0xff, col = (plan[j] >> 8) & 0xff;
      const int iy = ty0 - 1 + row, ix = tx0 - 1 + col;
      const bool inside = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W) & (row < IN_H);
      const int sy = a.ups2 ? (iy >> 1) : iy, sx = a.ups2 ? (ix >> 1) : ix;
      off[j] = inside ? (uint32_t)(tn * Hs + sy) * (uint32_t)Ws + (uint32_t)sx : OOB;
    }
  };
  // One wave-level DMA instruction (1 KB) into a ring slot, issued in four parts that are placed between
  // consecutive MFMAs (at one wave per SIMD a block of a dozen VALU in a row is a hole in the matrix pipe).
  // pA / pB: the chunk's two source planes (a missing plane = the zero page with record stride mul = 0, so the
  // instruction stream has no branch).
  struct DmaState { int dg; bool second, in; const char* bsel; uint32_t mul, off; const char* src; };
  DmaState ds{};
  auto dma_part = [&](int part, int j, const char* pA, const char* pB, uint32_t mulA, uint32_t mulB, const uint32_t (&off)[NDMA],
                      uint32_t fill_base) __attribute__((always_inline)) {
    if (part == 0) {
      // behind an opaque copy: the selects below depend only on (lane, j, chunk), and hoisted out of the
      // tile loop for every (j, chunk) pair they would hold 180 VGPRs
      int pj = plan[j];
      asm volatile("" : "+v"(pj));
      ds.dg = pj >> 16; ds.second = (ds.dg & 2) != 0; ds.off = off[j]; ds.in = ds.off != OOB;
    } else if (part == 1) {
      const char* base = ds.second ? pB : pA;
      ds.mul = ds.in ? (ds.second ? mulB : mulA) : 0u;
      ds.bsel = ds.in ? base : a.zero_page;
    } else if (part == 2) {
      ds.src = ds.bsel + (size_t)ds.off * ds.mul + (size_t)((ds.dg & 1) << 4);
    } else {
      dma16(ds.src, __builtin_amdgcn_readfirstlane(fill_base + j * (NW * 1024)));
    }
  };

  // tiles in flight: [0] the one being computed, [1..NT-1] the next ones the prefetch already reaches into
  // (three K-chunks ahead: the tile after next when a tile has only two chunks)
  constexpr int NT = (NCH + 2) / NCH + 1;
  int tl[NT], tn[NT], ty0[NT], tx0[NT];
  uint32_t offs[NT][NDMA];
  int kt = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    tl[t] = tile_of(t); tn[t] = ty0[t] = tx0[t] = 0;
    if (tl[t] >= 0) { setup_tile(tl[t], tn[t], ty0[t], tx0[t]); tile_offsets(tn[t], ty0[t], tx0[t], offs[t]); }
  }
  if (tl[0] < 0) return;
  auto plane_or_zero = [&](int j, bool on, const char*& ptr, uint32_t& mul) __attribute__((always_inline)) {
    const char* pl = on ? a.plane[j] : nullptr;
    ptr = pl ? pl : a.zero_page; mul = pl ? 32u : 0u;
  };

  // prologue: the first three K-chunks of the walk into slots 0, 1, 2
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int t = g / NCH, cc = g % NCH;
    const char *pA, *pB; uint32_t mA, mB;
    plane_or_zero(2 * cc, tl[t] >= 0, pA, mA); plane_or_zero(2 * cc + 1, tl[t] >= 0, pB, mB);
    const uint32_t fb = lds0 + g * STAGE + wave * 1024;
#pragma unroll
    for (int j = 0; j < NDMA; ++j)
#pragma unroll
      for (int part = 0; part < 4; ++part) dma_part(part, j, pA, pB, mA, mB, offs[t], fb);
  }
  wait_vm<2 * NDMA>();     // chunk 0 has landed (this wave's pieces of chunks 1 and 2 are still in flight)
  __builtin_amdgcn_s_barrier();
  int slot = 0;            // ring slot of the chunk being computed

  // residual through the matrix core (RL): A fragment of (1 / alpha) * I restricted to this wave's 16 channels
  // of x, which are half (cg & 1) of K-chunk cg >> 1:  A[m][k] = 1 / alpha  iff  k = 16 * (cg & 1) + m
  u32x4 a_res = {0u, 0u, 0u, 0u};
  if constexpr (RL) {
    const int m = lane & 15, kq = lane >> 4;
    if (kq == 2 * (cg & 1) + (m >> 3)) {
      const unsigned hv = (unsigned)__half_as_ushort(__float2half(1.f / a.alpha));
      a_res[(m & 7) >> 1] = hv << (16 * (m & 1));
    }
  }
  // bias as the C operand of each accumulator's first MFMA of a tile (no per-tile v_mov of 128 registers)
  f32x4 bias_q[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) bias_q[cb][i] = bias_v[4 * cb + i];

  // B fragments of NBF consecutive steps (read-ahead of LA = NBF - 1 steps, running across chunk and tile
  // boundaries; NBF divides the steps of a chunk so that the ring index of a step is static).  With four waves
  // reading and the DMA writing, an LDS read is back after 150-250 cycles: a step is only 96-192 cycles of MFMAs.
  constexpr int NBF = SS4K_RS_NBF;
  constexpr int LA = NBF - 1;
  static_assert(NSTEP % NBF == 0, "read-ahead ring must divide the steps of a chunk");
  u32x4 bf[NBF][2];
  auto ldb = [&](int s, const char* sbase) __attribute__((always_inline)) {   // s: step inside the chunk whose stage starts at sbase
    const int dx = s / (ROWS + 2), ir = s % (ROWS + 2);
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) bf[s % NBF][pb] = *reinterpret_cast<const u32x4*>(sbase + rdb[dx][pb] + ir * ROWB);
  };
#pragma unroll
  for (int s = 0; s < LA; ++s) ldb(s, smem);

  while (true) {
    f32x4 acc[ROWS][2][CB];
    const int n = tn[0], y0 = ty0[0], x0 = tx0[0];
    // epilogue addressing of this tile: lane (p, q) holds, per row mb and pixel block pb, channels
    // cout0 + NV*q .. + NV-1 of pixel (y0 + rg*ROWS + mb, x0 + 16*pb + p)
    const float alpha = a.alpha, gamma = a.gamma;
    int qe = q;
    asm volatile("" : "+v"(qe));   // re-derive the plane pointers per tile instead of carrying them through the loop
    const int opl = cout0 / 16 + (CB == 2 ? (qe >> 1) : 0);
    const size_t sub = CB == 2 ? (size_t)((qe & 1) << 4) : (size_t)(qe << 3);
    const size_t rec0 = (((size_t)n * a.H + y0 + rg * ROWS) * a.W + x0 + p) * 32;
    char* outp = a.out + (size_t)opl * a.out_plane_bytes + sub + rec0;
    const size_t row_bytes = (size_t)a.W * 32;
    const bool x_ok[2] = {x0 + p < a.W, x0 + 16 + p < a.W};
    typedef typename std::conditional<CB == 2, u32x4, u32x2>::type rvec;   // NV fp16 channels
    // activation, scaling and the fp16 store of one finished output row (no residual from memory)
    auto finish_row = [&](int mb) __attribute__((always_inline)) {
      if (y0 + rg * ROWS + mb >= a.H) return;   // wave-uniform
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        float v[NV];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = acc[mb][pb][cb][i], neg = t * slope_v[PR ? 4 * cb + i : 0];
            v[4 * cb + i] = (PR ? (t >= 0.f ? t : neg) : fmaxf(t, neg)) * (RL ? alpha * gamma : alpha);   // one slope in [0,1] unless PReLU
          }
        rvec o;
#pragma unroll
        for (int k = 0; k < NV; k += 2)
          o[k >> 1] = (unsigned)__half_as_ushort(__float2half(v[k])) | ((unsigned)__half_as_ushort(__float2half(v[k + 1])) << 16);
        if (x_ok[pb]) __builtin_nontemporal_store(o, reinterpret_cast<rvec*>(outp + mb * row_bytes + pb * 512));
      }
    };

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the stage filled during this chunk: three K-chunks ahead (of this tile or of a later one).  Past the
      // last tile the DMAs are still issued (zeros, into a free slot): the counted waits stay exact and the MFMA
      // stream carries no branch.
      const int tt = (c + 3) / NCH, cc = (c + 3) % NCH;
      const char *pA, *pB; uint32_t mulA, mulB;
      plane_or_zero(2 * cc, tl[tt] >= 0, pA, mulA); plane_or_zero(2 * cc + 1, tl[tt] >= 0, pB, mulB);
      const uint32_t fill_base = lds0 + ((slot + 3) & 3) * STAGE + wave * 1024;   // the slot read during the previous chunk
      const char* sb = smem + slot * STAGE;
      const char* sb_next = smem + ((slot + 1) & 3) * STAGE;
      constexpr int MID = NSTEP / 2;           // the chunk's one barrier sits here
      auto step = [&](int s) __attribute__((always_inline)) {
        const int dx = s / (ROWS + 2), ir = s % (ROWS + 2);
        if (s + LA < NSTEP) ldb(s + LA, sb); else ldb(s + LA - NSTEP, sb_next);   // next chunk's stage is visible since MID
        __builtin_amdgcn_sched_barrier(0);
        // DMA instructions of the prefetch: all after the barrier, spread evenly over the remaining steps
        // (DMA j at step MID + 1 + j * (steps left) / NDMA), each in four parts between consecutive MFMAs
        constexpr int LEFT = NSTEP - 1 - MID;
        const int dma_i = s <= MID ? 0 : ((s - 1 - MID) * NDMA + LEFT - 1) / LEFT;                    // issued before this step
        const int ndma_here = (s <= MID ? 0 : ((s - MID) * NDMA + LEFT - 1) / LEFT) - dma_i;          // 0, 1 or 2
        int part = 0, mcount = 0;
        // the interleaved epilogue: row ir-3 got its last MFMA one step ago (non-residual builds, last chunk, dx = 2)
        const bool fin_here = !RES && c == NCH - 1 && dx == 2 && ir >= 3;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int mb = ir - dy;
          if (mb >= 0 && mb < ROWS) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
              for (int cb = 0; cb < CB; ++cb) {
                if (c == 0 && dx == 0 && dy == 0) mfma16_init(acc[mb][pb][cb], W[c][dy * 3 + dx][cb], bf[s % NBF][pb], bias_q[cb], c < NA);
                else mfma16(acc[mb][pb][cb], W[c][dy * 3 + dx][cb], bf[s % NBF][pb], c < NA);
                if (part < 4 * ndma_here) {
                  __builtin_amdgcn_sched_barrier(0);
                  dma_part(part & 3, dma_i + (part >> 2), pA, pB, mulA, mulB, offs[tt], fill_base);
                  ++part;
                  __builtin_amdgcn_sched_barrier(0);
                }
                ++mcount;
                if (fin_here && mcount == 2) {
                  __builtin_amdgcn_sched_barrier(0);
                  asm volatile("s_nop 7");   // the row's last MFMA (end of the previous step) -> VALU read
                  finish_row(ir - 3);
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            if constexpr (RL) {
              // + x / alpha at the centre tap: one more MFMA per pixel block with the identity fragment
              if (c < 2 && dx == 1 && dy == 1 && (cg >> 1) == c) {
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) mfma16(acc[mb][pb][0], a_res, bf[s % NBF][pb], false);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k >= part && k < 4 * ndma_here) dma_part(k & 3, dma_i + (k >> 2), pA, pB, mulA, mulB, offs[tt], fill_base);
        if (fin_here && mcount < 2) { asm volatile("s_nop 7"); finish_row(ir - 3); }
        __builtin_amdgcn_sched_barrier(0);
      };
      // two fully unrolled halves with the chunk's one hand-over between them (a barrier inside a conditional
      // of the step loop would keep the loop from being unrolled)
#pragma unroll
      for (int s = 0; s <= MID; ++s) step(s);
      // the NEXT chunk's stage (issued two chunks ago) must have landed, the younger stage stays in flight - and
      // every wave is past the previous chunk, whose slot is refilled from here on.  vmcnt counts DMA, loads and
      // stores together in issue order: after a tile boundary the previous tile's output stores are older than the
      // stage left in flight, so this wait also covers them - half a chunk after they were issued, i.e. for free.
      wait_vm<NDMA>();
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int s = MID + 1; s < NSTEP; ++s) step(s);
      slot = (slot + 1) & 3;
    }

    if constexpr (!RES) {
      asm volatile("s_nop 11");   // last MFMA's result -> VALU read
      finish_row(ROWS - 1);
    } else {
      // ---------------- epilogue with residual(s) read from memory ----------------
      // fetched RB rows at a time (all loads of a batch first: one exposed round trip per batch, not per row -
      // nothing else runs on this SIMD).  hipcc waits for these loads with a vmcnt that also drains the ring.
      asm volatile("s_nop 11");   // last MFMA's result -> VALU read
      const char* r1p = (a.res1 && !RL) ? a.res1 + (size_t)opl * a.r1_plane_bytes + sub + rec0 : nullptr;
      const char* r2p = a.res2 ? a.res2 + (size_t)opl * a.r2_plane_bytes + sub + rec0 : nullptr;
      constexpr int RB = ROWS < 8 ? ROWS : 8;
#pragma unroll
      for (int mb0 = 0; mb0 < ROWS; mb0 += RB) {
        rvec u1[RB][2], u2[RB][2];
        const bool any_res = r1p || r2p;
        if (any_res) {
#pragma unroll
          for (int j = 0; j < RB; ++j)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb) {
              const bool ok = (y0 + rg * ROWS + mb0 + j < a.H) & x_ok[pb];
              // out-of-image lanes read the tile's first pixel instead (always inside) and are never stored
              const size_t ro = ok ? (size_t)(mb0 + j) * row_bytes + pb * 512 : (size_t)0 - (size_t)(rg * ROWS) * row_bytes - (size_t)p * 32;
              u1[j][pb] = *reinterpret_cast<const rvec*>((r1p ? r1p : r2p) + ro);
              u2[j][pb] = *reinterpret_cast<const rvec*>((r2p ? r2p : r1p) + ro);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < RB; ++j)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const int mb = mb0 + j;
            const bool ok = (y0 + rg * ROWS + mb < a.H) & x_ok[pb];
            float v[NV];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float t = acc[mb][pb][cb][i], neg = t * slope_v[PR ? 4 * cb + i : 0];
                v[4 * cb + i] = (PR ? (t >= 0.f ? t : neg) : fmaxf(t, neg)) * alpha;
              }
            if (any_res) {
              const rvec w1 = u1[j][pb], w2 = u2[j][pb];
              const float m1 = r1p ? 1.f : 0.f, m2 = r2p ? 1.f : 0.f;   // an absent residual is read as the other one, times 0
#pragma unroll
              for (int k = 0; k < NV; ++k) {
                const unsigned b1 = (w1[k >> 1] >> (16 * (k & 1))) & 0xffffu, b2 = (w2[k >> 1] >> (16 * (k & 1))) & 0xffffu;
                v[k] = (v[k] + m1 * __half2float(__ushort_as_half((unsigned short)b1))) * gamma + m2 * __half2float(__ushort_as_half((unsigned short)b2));
              }
            } else if (RL) {
#pragma unroll
              for (int k = 0; k < NV; ++k) v[k] *= gamma;
            }
            if (ok) {
              rvec o;
#pragma unroll
              for (int k = 0; k < NV; k += 2)
                o[k >> 1] = (unsigned)__half_as_ushort(__float2half(v[k])) | ((unsigned)__half_as_ushort(__float2half(v[k + 1])) << 16);
              __builtin_nontemporal_store(o, reinterpret_cast<rvec*>(outp + mb * row_bytes + pb * 512));
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (tl[1] < 0) break;
    // rotate the tile window
    ++kt;
#pragma unroll
    for (int t = 0; t + 1 < NT; ++t) {
      tl[t] = tl[t + 1]; tn[t] = tn[t + 1]; ty0[t] = ty0[t + 1]; tx0[t] = tx0[t + 1];
#pragma unroll
      for (int j = 0; j < NDMA; ++j) offs[t][j] = offs[t + 1][j];
    }
    tl[NT - 1] = tile_of(kt + NT - 1);
    if (tl[NT - 1] >= 0) { setup_tile(tl[NT - 1], tn[NT - 1], ty0[NT - 1], tx0[NT - 1]); tile_offsets(tn[NT - 1], ty0[NT - 1], tx0[NT - 1], offs[NT - 1]); }
  }
  wait_vm<0>();   // the trailing (zero) prefetches must have landed before the workgroup's LDS is released
}

template <int NCH, int ROWS, int CB, int CG, bool PR, bool RL = false, bool RES = false>
static void launch_t(ss4k_ctx* ctx, const ConvArgs& c, hipStream_t st) {
  SS4K_REQUIRE(RES || ((RL || !c.res1) && !c.res2), "internal: conv3x3_rs build without residual support");
  constexpr int RG = TH / ROWS, NW = RG * CG, COUT_WG = CG * CB * 16;
  constexpr size_t lds = (size_t)NSTAGE * STAGE;
  RsArgs a{};
  const int nplanes = c.nchunks0 + c.nchunks1;
  for (int j = 0; j < MAX_PLANES; ++j) {
    if (j < c.nchunks0) a.plane[j] = c.in0 + (size_t)(c.in0_plane0 + j) * c.in0_plane_bytes;
    else if (j < nplanes) a.plane[j] = c.in1 + (size_t)(c.in1_plane0 + j - c.nchunks0) * c.in1_plane_bytes;
    else a.plane[j] = nullptr;
  }
  a.zero_page = c.zero_page; a.wrs = c.wrs; a.bias = c.bias; a.prelu = c.prelu;
  a.out = c.out + (size_t)c.out_plane0 * c.out_plane_bytes; a.out_plane_bytes = c.out_plane_bytes;
  a.res1 = c.res1 ? c.res1 + (size_t)c.r1_plane0 * c.r1_plane_bytes : nullptr; a.r1_plane_bytes = c.r1_plane_bytes;
  a.res2 = c.res2 ? c.res2 + (size_t)c.r2_plane0 * c.r2_plane_bytes : nullptr; a.r2_plane_bytes = c.r2_plane_bytes;
  a.N = c.N; a.n0 = c.n0; a.H = c.H; a.W = c.W; a.ups2 = c.ups2; a.reverse = c.reverse; a.no_band = c.no_band;
  a.tiles_x = (c.W + TW - 1) / TW;
  a.tiles_y = (c.H + TH - 1) / TH;
  a.slope = c.act == ACT_LRELU ? c.slope : 1.f; a.alpha = c.alpha; a.gamma = c.gamma;
  const int groups = c.cout_pad / COUT_WG;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const void* fn = reinterpret_cast<const void*>(&conv3x3_rs_kernel<NCH, ROWS, CB, CG, PR, RL, RES>);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int gx = std::min(ntiles, std::max(1, (int)(ctx->num_cu / groups * (c.grid_share > 0.f ? c.grid_share : 1.f))));
  ctx->prof_family = "rs::conv3x3_rs_kernel (register-stationary weights, 16x16x32 MFMA)";
  hipLaunchKernelGGL((conv3x3_rs_kernel<NCH, ROWS, CB, CG, PR, RL, RES>), dim3(gx, groups), dim3(64 * NW), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace rs

// Layer shapes the register-stationary kernel is built for: (32-channel chunks, couts) -> <NCH, ROWS, CB, CG>
// (CG = cout groups; waves = (16 / ROWS) * CG).  wide: reserved (eight-wave variants, not built).
bool rs_config(int nplanes, int cout_pad, bool wide, int* nch, int* rows, int* cb, int* cg) {
  const int c = (nplanes + 1) / 2;
  (void)wide;
  if (cout_pad == 32 && c >= 2 && c <= 4) { *nch = c; *rows = 4; *cb = 2; *cg = 1; return true; }
  if (cout_pad == 32 && c == 5) { *nch = 5; *rows = 8; *cb = 1; *cg = 2; return true; }
  if (cout_pad == 64 && c == 2) { *nch = 2; *rows = 8; *cb = 2; *cg = 2; return true; }
  if (cout_pad == 64 && c == 6) { *nch = 6; *rows = 16; *cb = 1; *cg = 4; return true; }
  return false;
}

void launch_conv3x3_rs(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st) {
  int nch, rows, cb, cg;
  SS4K_REQUIRE(rs_config(a.nchunks0 + a.nchunks1, a.cout_pad, a.rs_wide != 0, &nch, &rows, &cb, &cg), "conv3x3_rs: unsupported layer shape");
  SS4K_REQUIRE(a.wrs && a.epi == EPI_NHWC && !a.bsvd_resid && a.act != ACT_RELU6, "conv3x3_rs: unsupported epilogue");
  SS4K_REQUIRE(a.act != ACT_PRELU || (rows == 8 && cb == 2 && !a.res1 && !a.res2), "conv3x3_rs: PReLU is built for the 64->64 shape without residuals only");
  SS4K_REQUIRE(a.act != ACT_LRELU || (a.slope >= 0.f && a.slope <= 1.f), "conv3x3_rs: LeakyReLU slope must be in [0,1]");
  const bool res = a.res1 || a.res2;
  SS4K_REQUIRE(!res || a.cout_pad == 64, "conv3x3_rs: residual epilogues are built for the 64-cout shapes only");
  if (rows == 4 && cg == 1) {
    if (nch == 2) rs::launch_t<2, 4, 2, 1, false>(ctx, a, st);
    else if (nch == 3) rs::launch_t<3, 4, 2, 1, false>(ctx, a, st);
    else rs::launch_t<4, 4, 2, 1, false>(ctx, a, st);
  } else if (rows == 8 && cb == 1) rs::launch_t<5, 8, 1, 2, false>(ctx, a, st);
  else if (rows == 8 && cb == 2) {
    if (a.act == ACT_PRELU) rs::launch_t<2, 8, 2, 2, true>(ctx, a, st);
    else if (res) rs::launch_t<2, 8, 2, 2, false, false, true>(ctx, a, st);
    else rs::launch_t<2, 8, 2, 2, false>(ctx, a, st);
  } else {
    // conv5 of an RDB: res1 is the conv's own input tensor (segment 0, 4 planes) and there is no activation
    const bool res_is_input = a.res1 && a.act == ACT_NONE && a.alpha != 0.f && a.nchunks0 == 4 && a.cout_pad == 64 &&
                              a.res1 + (size_t)a.r1_plane0 * a.r1_plane_bytes == a.in0 + (size_t)a.in0_plane0 * a.in0_plane_bytes &&
                              a.r1_plane_bytes == a.in0_plane_bytes && !a.ups2;
    if (res_is_input && !a.res2) rs::launch_t<6, 16, 1, 4, false, true, false>(ctx, a, st);
    else if (res_is_input) rs::launch_t<6, 16, 1, 4, false, true, true>(ctx, a, st);
    else if (res) rs::launch_t<6, 16, 1, 4, false, false, true>(ctx, a, st);
    else rs::launch_t<6, 16, 1, 4, false>(ctx, a, st);
  }
}

}  // namespace ss4k
