// Two chained 3x3 convolutions of a full-resolution, 32-channel-wide pair as ONE row-marching kernel: BSVD's inc
// (4|32 -> 30 -> 32, ReLU6 twice) and outc (32 -> 32 ReLU6 -> 32|3 with the denoiser's residual) blocks
// (reference bsvd/model.py:231-240, :318-323, :436-442).  Launched one after the other these layers are HBM-bound - 128 B of
// records per pixel and layer at 4.5 TB/s - and the 32-channel tensor between them is half of that traffic; here it lives in
// an LDS ring and never reaches HBM.
//
// A workgroup marches down a band of rows of a strip of 62 output columns.  Two waves are conv A (inter columns
// x0-1 .. x0+62 in two 32-pixel units), two are conv B (output columns x0 .. x0+61, two units, the last two columns of
// the second one discarded); a ring of 7 | 10 input rows (filled by LDS-DMA six | nine rows ahead) and a four-row ring of inter
// rows in LDS, one LDS-only barrier per row.  Arithmetic is the per-launch kernel's, operation for operation (conv_mfma.hip): fp32
// accumulators start from the bias, MFMAs in (K-chunk, dx, dy) order on the SAME packed weight fragments (pack.cpp: a lane's
// 16-byte fragment is read straight from the layer's blob into registers - 72 registers hold a 32 -> 32 layer), ReLU6, fp16
// round-to-nearest of the inter tensor - so the pair is BIT-IDENTICAL to the two launches it replaces.
// Records are the tensors' own: 16 channels = 32 bytes per pixel and plane; the two 16-byte halves of record p are swapped
// when ((p >> 2) ^ (p >> 3)) & 1, which makes every ds_read_b128 of a wave conflict-free at all three dx offsets.
#include "common.h"
#include "conv_tile.h"
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace ss4k {
namespace pair {

constexpr int OUTC = 62, RECS = 66, REC = 32, ROWB = RECS * REC;
// Input ring slots: a row is requested NS-1 steps before conv A first reads it and NS-4 rows' requests may be in flight across a
// barrier.  7 (two input planes) / 10 (one) are the most that keep three workgroups per CU; ten slots at two per CU were slower.
template <int PA> constexpr int pair_ns() { return PA == 2 ? 7 : 10; }
// roles are swapped on every other group of 2^PAIR_SWAP_SHIFT workgroups (see roleA)
#ifndef PAIR_SWAP_SHIFT
#define PAIR_SWAP_SHIFT 8
#endif
__device__ __forceinline__ int rec_off(int p, int h) { return p * REC + ((h ^ (((p >> 2) ^ (p >> 3)) & 1)) << 4); }

enum { EPI_RELU6 = 0, EPI_RESID = 1, EPI_RESID_NCHW = 2 };
// min(max(v, 0), 6) as one v_med3_f32 (the same value for every non-NaN v, and a NaN accumulator is not a result to preserve)
__device__ __forceinline__ float relu6(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 6.f); }

template <int PA, int EPI, bool STAMP = false>
__global__ __launch_bounds__(256, 3) void conv3x3_pair_kernel(const PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool RES = EPI != EPI_RELU6;
  constexpr int IN_SLOTB = PA * ROWB, MID_SLOTB = 2 * ROWB, NS = pair_ns<PA>(), AH = NS - 1;
  char* in_ring = smem;                        // [NS rows][PA planes][66 records]: rows t..t+2 being read, t+3..t+NS-1 landing
  char* mid_ring = smem + NS * IN_SLOTB;       // [4 rows][2 planes][66 records]
  char* res_ring = mid_ring + 4 * MID_SLOTB;   // [NS rows][64 pixels] x 16 bytes: channels 0..7 of the skip tensor (residual epilogues)
  constexpr int LDS_BYTES = NS * IN_SLOTB + 4 * MID_SLOTB + (RES ? NS * 1024 : 0);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the rings (DMA destination)
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), unit = wave & 1;
  // dev build, STAMP: cycles of this wave per phase (s_memtime): [0] DMA issue, [1] operand reads + MFMAs, [2] epilogue, [3] vmcnt wait, [4] barrier
  unsigned long long ph[5] = {0, 0, 0, 0, 0}, tlast = 0, rt0 = 0, ct0 = 0;
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
  // conv A on waves 0, 1 and conv B on 2, 3 - swapped on every other group of 256 workgroups: a workgroup's waves go to fixed SIMDs, and
  // with one assignment everywhere two SIMDs of every CU would carry all the conv A waves (measured: 3500 cycles per row there, the
  // conv B waves 2100 of them idle at the barrier)
  const bool roleA = (wave < 2) != (((blockIdx.x >> PAIR_SWAP_SHIFT) & 1) != 0);
  const int strips = (a.W + OUTC - 1) / OUTC;
  const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) % a.bands, frame = a.n0 + blockIdx.x / (strips * a.bands);
  const int rpb = (a.H + a.bands - 1) / a.bands, ylo = band * rpb, yhi = min(a.H, ylo + rpb);
  if (frame >= a.n0 + a.N || ylo >= yhi) return;
  const int x0 = strip * OUTC;
  for (int e = tid; e < LDS_BYTES / 16; e += 256) reinterpret_cast<uint4*>(smem)[e] = make_uint4(0u, 0u, 0u, 0u);

  // this wave's layer: weight fragments [(chunk*3 + dx)*3 + dy] and the bias of the channels its accumulators hold
  uint4 Wt[18];
  {
    const char* wsrc = (roleA ? a.wA : a.wB) + lane * 16;
    const int nfr = roleA ? 9 * PA : 18;
#pragma unroll
    for (int i = 0; i < 18; ++i) Wt[i] = i < nfr ? *reinterpret_cast<const uint4*>(wsrc + i * 1024) : make_uint4(0u, 0u, 0u, 0u);
  }
  // the bias of the channels a lane's accumulator holds (element i = channel 16 (i >> 3) + 8 h + (i & 7)): both halves' values are wave-uniform
  // (scalar registers), a lane selects its half's per row - held as sixteen vector registers through the march three of the six builds
  // spilled 6-10 registers at the 168 of three workgroups per CU
  float bias_s[2][16];
  {
    const float* bs = roleA ? a.biasA : a.biasB;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int i = 0; i < 16; ++i) bias_s[hh][i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bs[16 * (i >> 3) + 8 * hh + (i & 7)])));
  }
  auto acc_init = [&](f32x16& acc) {
    int hh = h;
    asm volatile("" : "+v"(hh));   // selected per row: hoisted out of the march the sixteen values are vector registers again
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = hh ? bias_s[1][i] : bias_s[0][i];
  };
  // pixel operand of tap dx: record 32*unit + n + dx of the ring row (conv A: input column x0-2 + that; conv B: inter
  // column x0-1 + that), 16-byte half h
  int rd[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) rd[dx] = rec_off(32 * unit + n + dx, h);
  const int pcol = 32 * unit + n;                 // this lane's pixel: inter record (conv A) / output column x0 + pcol (conv B)
  const int wr = rec_off(pcol, h);
  const int xi = x0 - 1 + pcol;                   // conv A: image column of the inter pixel
  const bool a_col_in = xi >= 0 && xi < a.W;
  const int xo = x0 + pcol;                       // conv B: output column
  const bool b_ok = pcol < OUTC && xo < a.W;

  // Input rows (and the skip tensor's) arrive by LDS-DMA, issued NS-1 rows ahead and awaited with a COUNTED
  // s_waitcnt (NS-4 rows' worth may stay in flight across the barrier).  Loads through registers did not work here: hipcc waits with
  // vmcnt(0) for any load under a branch or whose register is copied, and __syncthreads() drains the counter too - either turns the
  // prefetch into a round trip to HBM per row (measured: 2.0-2.6 us per row against 0.8 us of MFMA time).
  // A row is 132*PA 16-byte pieces, contiguous in the ring (piece e = plane*132 + record*2 + physical half): DMA instruction k moves
  // pieces 64k..64k+63, lane by lane; the conv B waves issue them (they have the slack: the conv A waves carry the fp16 conversion and the
  // LDS stores of the inter row), one the even k, the other the odd k and the skip row.  Zero padding = the zero page.
  constexpr int NPIECE = 132 * PA, NK = (NPIECE + 63) / 64, MAXD = 3;
  size_t dsrc[MAXD]; bool dok[MAXD]; int dk[MAXD];   // this wave's DMA instructions of one row: source (column part), lane valid, k (-1: none, NK: skip row)
  int nd = 0;   // how many of them exist (wave-uniform)
#pragma unroll
  for (int i = 0; i < MAXD; ++i) {
    const int k = 2 * i + unit;   // conv B waves only (the others never issue)
    dk[i] = -1; dok[i] = false; dsrc[i] = 0;
    if (!roleA && k < NK) {
      const int e = 64 * k + lane, ec = min(e, NPIECE - 1), pl = ec / 132, rem = ec - pl * 132, rec = rem >> 1;
      const int hf = (rem & 1) ^ (((rec >> 2) ^ (rec >> 3)) & 1);   // the logical half that sits at this physical position (rec_off)
      const int x = x0 - 2 + rec;
      dk[i] = k; dok[i] = e < NPIECE && x >= 0 && x < a.W;
      dsrc[i] = (size_t)(a.in_plane0 + pl) * a.in_plane_bytes + ((size_t)frame * a.H * a.W + min(max(x, 0), a.W - 1)) * REC + 16 * hf;
      nd = i + 1;
    }
  }
  if (RES && !roleA && unit == 1) {   // the skip row: lane = pixel x0 + lane, channels 0..7
    dk[nd] = NK; dok[nd] = true;
    dsrc[nd] = (size_t)a.res_plane0 * a.res_plane_bytes + ((size_t)frame * a.H * a.W + min(x0 + lane, a.W - 1)) * REC;
    ++nd;
  }
  nd = __builtin_amdgcn_readfirstlane(nd);
  const int npiece_last = NPIECE - 64 * (NK - 1);   // lanes of the last row instruction that carry a piece
  // issue the DMA bundle of input row r (image row ylo-2+r) into slot sl_in, and the skip row of step r-2 (image row ylo+r-5) into sl_res
  const size_t row_bytes = (size_t)a.W * REC;
  const char* rowsrc[MAXD];   // source of input row r (advanced by one image row per bundle; rows outside the image are never dereferenced)
#pragma unroll
  for (int i = 0; i < MAXD; ++i) rowsrc[i] = a.in + dsrc[i] + (ptrdiff_t)(ylo - 2) * (ptrdiff_t)row_bytes;
  auto issue = [&](int r, int sl_in, int sl_res) {   // called with r = 0, 1, 2, ... in order
    const int y = ylo - 2 + r, yr = min(max(ylo + r - 5, 0), a.H - 1);
    const bool row_in = y >= 0 && y < a.H;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
      if (i < nd) {   // wave-uniform
        if (dk[i] == NK) {
          dma16(a.res + dsrc[i] + (size_t)yr * row_bytes, __builtin_amdgcn_readfirstlane(lds0 + NS * IN_SLOTB + 4 * MID_SLOTB + sl_res * 1024));
        } else {
          const char* p = (row_in && dok[i]) ? rowsrc[i] : a.zero_page;
          rowsrc[i] += row_bytes;
          const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + sl_in * IN_SLOTB + dk[i] * 1024);
          if (dk[i] < NK - 1 || lane < npiece_last) dma16(p, dst);   // the last instruction of a row is partly filled: masked lanes move nothing
        }
      }
    }
  };
  auto wait_bundles = [&]() {   // all but the AH-3 newest bundles of this wave have landed
    constexpr int K = AH - 3;
    static_assert(3 * K <= 63, "vmcnt field");
    if (nd == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K) : "memory");
    else if (nd == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory");
    else if (nd == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
  };
  auto wrap = [](int v) { return v >= NS ? v - NS : v; };
  __syncthreads();   // rings are zero (and every zero store has landed before the first DMA may overwrite it)
  if (!roleA) {
    for (int r = 0; r < AH; ++r) issue(r, r, wrap(r + NS - 2));   // skip row of step r-2 -> slot (r-2) mod NS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const size_t plane_px = (size_t)a.H * a.W;
  const int rows = yhi - ylo;
  const int nsteps = rows + 3;   // conv A: inter row t (image row ylo-1+t) at step t <= rows+1; conv B: output row t-3 at step t
  // one copy of the loop per role (a wave takes exactly one)
  auto march = [&](auto role_a) {
  constexpr bool ROLE_A = decltype(role_a)::value;
  int s6 = 0;   // t mod NS: the ring slot of input row t and of the skip row of step t
  for (int t0 = 0; t0 < nsteps; t0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u;
      if (t >= nsteps) break;   // uniform over the workgroup
      stamp(-1);
      if constexpr (ROLE_A) {
        if (t <= rows + 1) {
          const int yq = ylo - 1 + t;
          uint4 o0 = make_uint4(0u, 0u, 0u, 0u), o1 = o0;
          if (yq >= 0 && yq < a.H) {   // wave-uniform; a row outside the image is conv B's zero padding
            f32x16 acc;
            acc_init(acc);
            const char* rowp[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rowp[dy] = in_ring + wrap(s6 + dy) * IN_SLOTB;
            // the nine pixel operands of a K-chunk are read together, then its nine MFMAs issue (left to itself hipcc re-uses ONE
            // register quad: read, wait, MFMA, eighteen times - 2.2 us per row); the next chunk's reads overlap this one's MFMAs
#pragma unroll
            for (int c = 0; c < PA; ++c) {
              uint4 b[9];
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) b[dx * 3 + dy] = *reinterpret_cast<const uint4*>(rowp[dy] + c * ROWB + rd[dx]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int i = 0; i < 9; ++i) acc = mma<__half>(Wt[c * 9 + i], b[i], acc);
              __builtin_amdgcn_sched_barrier(0);
            }
            stamp(1);
            __half* h0 = reinterpret_cast<__half*>(&o0); __half* h1 = reinterpret_cast<__half*>(&o1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              h0[i] = __float2half(a_col_in ? relu6(acc[i]) : 0.f);       // a column outside the image: padding too
              h1[i] = __float2half(a_col_in ? relu6(acc[8 + i]) : 0.f);
            }
          }
          char* dst = mid_ring + u * MID_SLOTB + wr;
          *reinterpret_cast<uint4*>(dst) = o0;
          *reinterpret_cast<uint4*>(dst + ROWB) = o1;
        }
      } else {
        issue(t + AH, wrap(s6 + AH), wrap(s6 + AH - 2));   // input row t+AH; skip row of step t+AH-2
        stamp(0);
        const int P = t - 3, y = ylo + P;
        if (P >= 0 && y < yhi) {
          f32x16 acc;
          acc_init(acc);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            uint4 b[9];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int dy = 0; dy < 3; ++dy)
                b[dx * 3 + dy] = *reinterpret_cast<const uint4*>(mid_ring + ((u + 1 + dy) & 3) * MID_SLOTB + c * ROWB + rd[dx]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 9; ++i) acc = mma<__half>(Wt[c * 9 + i], b[i], acc);
            __builtin_amdgcn_sched_barrier(0);
          }
          stamp(1);
          if (b_ok) {
            const size_t ipix = ((size_t)frame * a.H + y) * a.W + xo;
            float v0[8], v1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v0[i] = acc[i]; v1[i] = acc[8 + i]; }
            if constexpr (EPI == EPI_RELU6) {
#pragma unroll
              for (int i = 0; i < 8; ++i) { v0[i] = relu6(v0[i]); v1[i] = relu6(v1[i]); }
            } else if (h == 0) {   // channels 0..2: skip - conv (bsvd/model.py:436-442); the others pass through
              const uint2 rr = *reinterpret_cast<const uint2*>(res_ring + s6 * 1024 + pcol * 16);
              const __half* rh = reinterpret_cast<const __half*>(&rr);
#pragma unroll
              for (int i = 0; i < 3; ++i) v0[i] = __half2float(rh[i]) - v0[i];
            }
            if constexpr (EPI == EPI_RESID_NCHW) {
              if (h == 0) {
                float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                  if (i < a.cout_real) o[((size_t)frame * a.cout_real + i) * plane_px + (size_t)y * a.W + xo] = v0[i];
              }
            } else {
              char* o = a.out + (size_t)a.out_plane0 * a.out_plane_bytes + ipix * REC + 16 * h;
              store8<__half>(o, v0);
              store8<__half>(o + a.out_plane_bytes, v1);
            }
          }
        }
        stamp(2);
        wait_bundles();   // this wave's stores count too: they only make the wait stricter
        stamp(3);
      }
      s6 = wrap(s6 + 1);
      stamp(2);
      lds_barrier();   // not __syncthreads(): DMA bundles (and conv B's stores) stay in flight across it
      stamp(4);
    }
  }
  };
  if (roleA) march(std::true_type{}); else march(std::false_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may land after the workgroup has given its LDS back
  if constexpr (STAMP) {
    if (lane == 0 && a.dbg_buf && blockIdx.x < 1024) {
      unsigned long long* o = a.dbg_buf + ((size_t)blockIdx.x * 4 + wave) * 8;
      for (int k = 0; k < 5; ++k) o[k] = ph[k];
      o[5] = (unsigned long long)nsteps; o[6] = roleA ? 1 : 0;
      o[7] = ((__builtin_amdgcn_s_memtime() - ct0) << 20) / (__builtin_amdgcn_s_memrealtime() - rt0 + 1);   // shader cycles per 100 MHz tick, x 2^20
    }
  }
}

template <int PA, int EPI>
static void launch_t(ss4k_ctx* ctx, const PairArgs& a0, hipStream_t st) {
  constexpr size_t lds = (size_t)(pair_ns<PA>() * PA + 8) * ROWB + (EPI != EPI_RELU6 ? pair_ns<PA>() * 1024 : 0);
  constexpr int per_cu = (int)std::min<size_t>(3, 160 * 1024 / lds);
  static_assert(per_cu >= 2, "LDS budget");
  PairArgs a = a0;
  const int strips = (a.W + OUTC - 1) / OUTC;
  // one round of workgroups at two or three per CU (times this launch's share of the chip); a band re-does 4 input rows and 3 steps
  const int slots = std::max(1, (int)(per_cu * ctx->num_cu * (a.grid_share > 0.f ? a.grid_share : 1.f)));
  a.bands = std::max(1, std::min((a.H + 15) / 16, slots / std::max(1, a.N * strips)));
#ifdef SS4K_DEV
  static const bool stamp_mode = std::getenv("SS4K_PAIR_STAMP") && std::getenv("SS4K_PAIR_STAMP")[0] == '1';
  if (stamp_mode) {   // phase cycle counters of every wave of the first 1024 workgroups, printed per role
    static unsigned long long* dbuf = nullptr;
    if (!dbuf) SS4K_HIP(hipMalloc(reinterpret_cast<void**>(&dbuf), 1024 * 4 * 8 * 8));
    SS4K_HIP(hipMemsetAsync(dbuf, 0, 1024 * 4 * 8 * 8, st));
    a.dbg_buf = dbuf;
    const void* fs = reinterpret_cast<const void*>(&conv3x3_pair_kernel<PA, EPI, true>);
    if (ctx->lds_attr_set.insert(fs).second) SS4K_HIP(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((conv3x3_pair_kernel<PA, EPI, true>), dim3((unsigned)(a.N * a.bands * strips)), dim3(256), lds, st, a);
    SS4K_HIP(hipStreamSynchronize(st));
    std::vector<unsigned long long> hb(1024 * 4 * 8);
    SS4K_HIP(hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost));
    double acc[2][5] = {{0}}; double steps[2] = {0, 0}; double clk = 0; int nclk = 0;
    for (int wg = 0; wg < 1024; ++wg) for (int w = 0; w < 4; ++w) {
      const unsigned long long* o = &hb[((size_t)wg * 4 + w) * 8];
      if (!o[5]) continue;
      const int r = o[6] ? 0 : 1;
      clk += (double)o[7] / 1048576.0 * 100.0; ++nclk;
      for (int k = 0; k < 5; ++k) acc[r][k] += (double)o[k];
      steps[r] += (double)o[5];
    }
    if (nclk) std::fprintf(stderr, "[pair<%d,%d>] shader clock %.0f MHz (s_memtime / s_memrealtime)\n", PA, EPI, clk / nclk);
    for (int r = 0; r < 2; ++r) if (steps[r] > 0)
      std::fprintf(stderr, "[pair<%d,%d> %s] shader cycles per row (s_memtime): dma issue %.0f  reads+mfma %.0f  epilogue %.0f  vmcnt wait %.0f  barrier %.0f\n",
                   PA, EPI, r == 0 ? "conv A waves" : "conv B waves", acc[r][0] / steps[r], acc[r][1] / steps[r], acc[r][2] / steps[r], acc[r][3] / steps[r], acc[r][4] / steps[r]);
    return;
  }
#endif
  const void* fn = reinterpret_cast<const void*>(&conv3x3_pair_kernel<PA, EPI>);
  if (ctx->lds_attr_set.insert(fn).second) SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  ctx->prof_family = "pair::conv3x3_pair_kernel (two 32-channel full-resolution layers, row-marching, 32x32x16 MFMA: BSVD inc / outc)";
  hipLaunchKernelGGL((conv3x3_pair_kernel<PA, EPI>), dim3((unsigned)(a.N * a.bands * strips)), dim3(256), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace pair

bool conv3x3_pair_eligible(int planes_a, int cout_pad_a, int nchunks_b, int cout_pad_b) {
  return (planes_a == 1 || planes_a == 2) && cout_pad_a == 32 && nchunks_b == 2 && cout_pad_b == 32;
}

void launch_conv3x3_pair(ss4k_ctx* ctx, const PairArgs& a, hipStream_t st) {
  using namespace pair;
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && (a.planes_a == 1 || a.planes_a == 2), "conv3x3_pair: shape");
  SS4K_REQUIRE(a.epi == 0 || a.res, "conv3x3_pair: the residual epilogues need the skip tensor");
  SS4K_REQUIRE(a.epi != 2 || (a.cout_real >= 1 && a.cout_real <= 8), "conv3x3_pair: NCHW output takes at most 8 channels");
  ProfScope prof(ctx, st, PROF_CONV);
#define SS4K_PAIR(PA_) \
  switch (a.epi) { \
    case 0: launch_t<PA_, EPI_RELU6>(ctx, a, st); break; \
    case 1: launch_t<PA_, EPI_RESID>(ctx, a, st); break; \
    default: launch_t<PA_, EPI_RESID_NCHW>(ctx, a, st); break; \
  }
  if (a.planes_a == 1) { SS4K_PAIR(1) } else { SS4K_PAIR(2) }
#undef SS4K_PAIR
  prof.done(a.flops);
}

}  // namespace ss4k
