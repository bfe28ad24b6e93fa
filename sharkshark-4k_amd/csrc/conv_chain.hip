// Cross-layer execution of the RRDB body: a CHAIN of 3x3 convs (fp16, plain epilogue, 32-cout groups) as ONE persistent
// launch, with per-tile ready counters in place of the kernel boundary between dependent layers (profiles/NOTES_r01_r03.md 4.1d).
//
// Why: a 1-frame 720p job is 345 dependent launches of 360-460 tiles on 512 workgroup slots - every launch pays its boundary,
// the prologue of its workgroups (first K-chunk in flight, nothing to compute) and a partly filled round of tiles, and frame
// lanes cannot help (one frame).  tools/micro/chain_skeleton.hip measured the hand-off below at -10 % per layer against
// back-to-back launches on this dataflow, every word checked on reused buffers.
//
// Structure.  The chain is a list of ITEMS (one conv layer x one 32-cout group; conv5 of an RDB is two items); a work UNIT is
// (item, tile).  Units are handed out IN ORDER from one device-wide queue (an atomic ticket counter), so every unit that was
// handed out is held by a workgroup that is running: progress never depends on how many workgroups of the grid are resident
// (another stream's kernels, another process) - no co-residency assumption, no grid barrier.  A unit of layer L on tile t
//   * may read planes written by layers <= L-2 once all units of those layers have finished on t's 3 x 3 tile neighbourhood
//     (need_old), and the planes layer L-1 wrote once its units have (need_new).  The K loop walks the planes oldest first, so
//     the poll for need_new sits two chunks before the first NEWEST chunk is read and hides under the older chunks' MFMAs;
//   * may write once need_new holds: every reader of the region it overwrites (buffers are reused from RDB to RDB) belongs to
//     an earlier layer and to the neighbourhood.  Each layer reads the previous one's output, so by induction "layer L-1
//     finished on N(t)" implies every older layer finished on N(N(t)): one counter per tile is enough.
// Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility; validated adversarially by the skeleton): outputs are
// stored write-through (sc1, whole 1 KB runs per wave instruction); every storing wave drains (vmcnt(0)), the workgroup
// meets at a barrier, ONE lane adds 1 to the tile's counter (agent-scope atomic).  Consumers poll the 9 counters of the
// neighbourhood with relaxed agent-scope loads from one wave, a barrier releases the others, and EVERY load of handed-off
// bytes bypasses the CU's L1: the halo tiles come in by LDS-DMA with sc1 (the skeleton's "warm" test: 0 stale words with it,
// 4e5 without), residuals by sc1 buffer loads.  The counter add of a unit is deferred to the first chunk barrier of the
// workgroup's next unit - the stores drain under that chunk's MFMAs instead of in front of a wait - unless the workgroup is
// about to block, in which case it publishes first (a blocked workgroup never holds back a finished tile).
// The tile body (LDS image, swizzle, MFMA order, epilogue arithmetic) is conv_mfma.hip's <__half, 1, MB, 4> build: results are
// bit-identical to the one-launch-per-layer path on the LDS-weights kernel (tests/test_gpu_chain.py).
#include "common.h"
#include "conv_tile.h"

namespace ss4k {
namespace chain {

typedef __attribute__((address_space(1))) unsigned gu32;
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

constexpr int NW = 4;
constexpr unsigned SPIN_LIMIT = 1u << 21;   // polls (~1 us each) before a unit gives up: the kernel always drains

template <int MB> constexpr size_t lds_bytes_chain() {
  return (size_t)(2 * Geo<__half, MB, NW>::TILE_SLOTS + 2 * 9 * 64) * 16 + 2 * 64 * 4 + 16;
}

__device__ __forceinline__ void dma16_sc1(const void* gsrc, uint32_t lds_addr_wave_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr_wave_uniform)
               : "memory");
}

// a wave-uniform pointer as a buffer resource (raw, 2 GB window): stores / loads through it carry the cache bits as `aux`
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const char* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
constexpr int AUX_SC1 = 16;

// A unit's parameters, copied ONCE per unit from the item table into registers (scalar loads from the constant address space:
// the table is not written during the launch).  Reading an item field where it is used would put a load and its wait - which
// also drains the LDS-DMA prefetch and the previous unit's stores - into the chunk loop (measured: -10 % on the whole chain).
typedef const __attribute__((address_space(4))) ChainItem* CItem;
struct UnitP {
  const char* in0; size_t in0_pb; int nch0;     // first plane of segment 0, bytes per plane, chunks in segment 0
  const char* in1; size_t in1_pb; int nch;      // first plane of segment 1; total chunks
  const char* wpk; const float* bias;
  float slope, alpha, gamma;                    // slope 1 = no activation
  const char* res1; size_t r1_pb; const char* res2; size_t r2_pb;   // first plane of this group's residual planes or null
  char* out; size_t out_pb;
  int newest; unsigned need0, need_new;         // need0: what the first chunk waits for
  unsigned pub_need;
};
__device__ __forceinline__ UnitP load_unit(const ChainItem* items, int item) {
  CItem it = (CItem)(items + item);
  UnitP u;
  u.in0_pb = it->in0_plane_bytes; u.in0 = it->in0 + (size_t)it->in0_plane0 * u.in0_pb; u.nch0 = it->nchunks0;
  u.in1_pb = it->in1_plane_bytes; u.in1 = it->in1 ? it->in1 + (size_t)it->in1_plane0 * u.in1_pb : nullptr; u.nch = it->nchunks0 + it->nchunks1;
  u.wpk = it->wpk; u.bias = it->bias;
  u.slope = it->act == ACT_LRELU ? it->slope : 1.f; u.alpha = it->alpha; u.gamma = it->gamma;
  u.r1_pb = it->r1_plane_bytes; u.res1 = it->res1 ? it->res1 + (size_t)it->r1_plane0 * u.r1_pb : nullptr;
  u.r2_pb = it->r2_plane_bytes; u.res2 = it->res2 ? it->res2 + (size_t)it->r2_plane0 * u.r2_pb : nullptr;
  u.out_pb = it->out_plane_bytes; u.out = it->out + (size_t)it->out_plane0 * u.out_pb;
  u.newest = it->newest; u.need_new = it->need_new; u.need0 = it->newest == 0 ? it->need_new : it->need_old;
  u.pub_need = it->pub_need;
  return u;
}

template <int AUX>
__device__ __forceinline__ void store8_aux(__amdgpu_buffer_rsrc_t r, uint32_t voff, const float* v) {
  uint4 a;
  __half* ha = reinterpret_cast<__half*>(&a);
#pragma unroll
  for (int i = 0; i < 8; ++i) ha[i] = __float2half(v[i]);
  __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<u32x4v*>(&a), r, voff, 0, AUX);
}

// ABL: timing-only ablations of the dev library (results are void): 1 = plain LDS-DMA for the activations, 2 = non-temporal
// instead of write-through stores, 4 = no dependency polls at all
template <int MB> constexpr int chain_wgs_per_cu() { return lds_bytes_chain<MB>() * 3 <= 160 * 1024 ? 3 : 2; }

template <int MB, int ABL>
__global__ __launch_bounds__(64 * NW, chain_wgs_per_cu<MB>()) void conv3x3_chain_kernel(const ChainArgs ca) {
  auto store8_sc1 = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, const float* v) { if (ABL & 2) store8_aux<2>(r, voff, v); else store8_aux<AUX_SC1>(r, voff, v); };
  using T = __half;
  using G = Geo<T, MB, NW>;
  constexpr int SPR = G::SPR, REC = G::REC, NG = 3;
  constexpr int TH = G::TH, TILE_SLOTS = G::TILE_SLOTS, TILE_DMA = G::TILE_DMA, DMA_PER_WAVE = G::DMA_PER_WAVE;
  constexpr int WSLOTS = 9 * 64;
  constexpr int TILE_BYTES = TILE_SLOTS * 16, W_BYTES = WSLOTS * 16;
  static_assert(SPR == 2 && REC == 32, "fp16 records");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [tile buf 0][tile buf 1][weights buf 0][weights buf 1][epilogue constants: 2 parities x (32 bias + 32 slopes)][control words]
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  float* epi_lds = reinterpret_cast<float*>(smem + 2 * TILE_BYTES + 2 * W_BYTES);
  volatile unsigned* ctl = reinterpret_cast<volatile unsigned*>(epi_lds + 128);   // [0] next ticket, [1] next unit's first chunk may be read

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int tiles_per_frame = ca.tiles_y * ca.tiles_x;
  const int ntiles = ca.N * tiles_per_frame;
  const unsigned nwork = (unsigned)ca.nitems * (unsigned)ntiles;
  gu32* head = (gu32*)ca.ctl;
  gu32* errw = head + 1;
  // a unit that gives up marks the launch (errw: per launch, makes every other wait end so the grid drains) AND the model's sticky
  // word in pinned host memory, which no launch ever resets - only the host code that reports the error clears it
  auto raise_err = [&]() {
    __hip_atomic_fetch_or(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ca.err_sticky) __hip_atomic_fetch_or((gu32*)ca.err_sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  gu32* flags = head + 4;
#ifdef SS4K_DEV
  const unsigned spin_limit = ca.spin_limit ? ca.spin_limit : SPIN_LIMIT;
#else
  constexpr unsigned spin_limit = SPIN_LIMIT;
#endif

  auto swz = [](int x) { return (x >> 3) & 1; };
  int rd_base[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int x = lr + dx;
    rd_base[dx] = (((wave * MB) * IN_W + x) * SPR + (lh ^ swz(x))) * 16;
  }
  int plan[DMA_PER_WAVE];  // row | x << 8 | group*16 << 16, -1 = no slot
#pragma unroll
  for (int j = 0; j < DMA_PER_WAVE; ++j) {
    const int s = (wave + NW * j) * 64 + lane;
    const int p = s / SPR, gq = s % SPR;
    const int row = p / IN_W, x = p - row * IN_W;
    plan[j] = (s < TILE_SLOTS) ? (row | (x << 8) | (((gq ^ swz(x)) * 16) << 16)) : -1;
  }
  uint32_t src_off[DMA_PER_WAVE];
  auto setup_tile = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % ca.tiles_x, tyn = tile / ca.tiles_x;
    const int ty = tyn % ca.tiles_y;
    n = ca.n0 + tyn / ca.tiles_y; y0 = ty * TH; x0 = tx * TW;
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
      const int iy = y0 - 1 + (plan[j] & 0xff), ix = x0 - 1 + ((plan[j] >> 8) & 0xff);
      const bool ok = plan[j] >= 0 && iy >= 0 && iy < ca.H && ix >= 0 && ix < ca.W;
      src_off[j] = ok ? (uint32_t)(n * ca.H + iy) * (uint32_t)ca.W + (uint32_t)ix : OOB;
    }
  };
  // counter index of the 3 x 3 neighbourhood member this lane polls (lanes 0-8; -1: outside the frame / other lanes)
  auto nb_index = [&](int tile) -> int {
    if (lane >= 9) return -1;
    const int f = tile / tiles_per_frame, r = tile - f * tiles_per_frame;
    const int ty = r / ca.tiles_x + lane / 3 - 1, tx = r % ca.tiles_x + lane % 3 - 1;
    return (ty >= 0 && ty < ca.tiles_y && tx >= 0 && tx < ca.tiles_x) ? f * tiles_per_frame + ty * ca.tiles_x + tx : -1;
  };
  auto satisfied = [&](unsigned v, int nbi, unsigned need) -> bool { return __all(nbi < 0 || (int)(v - need) >= 0) != 0; };
  unsigned st_block = 0, st_spin0 = 0, st_spin1 = 0;   // ABL & 8 (dev): blocking starts, polls spent waiting at a unit's start / inside
  bool st_mid = false;
  auto poll_block = [&](int nbi, unsigned need) {   // one wave; returns when the neighbourhood has reached `need`
    if (ABL & 4) return;
    unsigned spins = 0;
    const unsigned long long t0 = (ABL & 8) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz
    while (true) {
      unsigned v = need;
      if (nbi >= 0) v = __hip_atomic_load(flags + nbi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (satisfied(v, nbi, need)) break;
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      // give up after SPIN_LIMIT polls - or as soon as any unit has (the error word is sticky: once a unit timed out the results
      // of this forward are void and every wait ends, so the launch drains in milliseconds); the host sees the error word
      const bool dead = (spins & 1023u) == 0 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (spins > spin_limit || dead) {
        if (lane == 0) raise_err();
        break;
      }
    }
    if (ABL & 8) { const unsigned dt = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0); if (st_mid) st_spin1 += dt; else st_spin0 += dt; }
  };

  constexpr int NDMA_T = DMA_PER_WAVE, NDMA_W = (9 + NW - 1) / NW, NDMA = NDMA_T + NDMA_W;
  constexpr int SE = NDMA <= NG * MB ? 3 : (2 * NDMA <= NG * 3 * MB ? 2 : 1);
  static_assert((3 * MB) % SE == 0 && NDMA <= NG * 3 * MB / SE, "not enough DMA slots in the MFMA stream");
  const char* pf_plane = nullptr; const char* pf_wsrc = nullptr;
  uint32_t pf_tdst = 0, pf_wdst = 0; bool pf_on = false;
  auto prefetch_begin = [&](const UnitP& u, int c, int buf) {
    pf_plane = (c < u.nch0) ? u.in0 + (size_t)c * u.in0_pb : u.in1 + (size_t)(c - u.nch0) * u.in1_pb;
    pf_tdst = lds0 + buf * TILE_BYTES;
    pf_wsrc = u.wpk + (size_t)c * W_BYTES + lane * 16;
    pf_wdst = lds0 + 2 * TILE_BYTES + buf * W_BYTES;
    pf_on = true;
  };
  auto dma_op = [&](int idx) {
    if (!pf_on) return;
    if (idx < NDMA_T) {
      const int k = wave + NW * idx;
      if (k < TILE_DMA) {
        const size_t boff = (size_t)src_off[idx] * REC + (size_t)((plan[idx] >> 16) & 0xff);
        const char* src = src_off[idx] != OOB ? pf_plane + boff : ca.zero_page + (lane & 3) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(pf_tdst + k * 1024);
        if (plan[idx] >= 0) { if (ABL & 1) dma16(src, dst); else dma16_sc1(src, dst); }   // activations: written by other workgroups of this launch
      }
    } else if (idx < NDMA) {
      const int k = wave + NW * (idx - NDMA_T);
      if (k < 9) dma16(pf_wsrc + k * 1024, __builtin_amdgcn_readfirstlane(pf_wdst + k * 1024));   // weights: never written here
    }
  };
  auto write_epi = [&](int par, const UnitP& u) {
    if (tid < 32) {
      epi_lds[par * 64 + tid] = u.bias[tid];
      epi_lds[par * 64 + 32 + tid] = u.slope;
    }
  };
  // The counter add of a finished unit: after every wave's vmcnt(0) (its write-through stores have left) and a workgroup
  // barrier.  (Measured and dropped: every wave publishing for itself right after its epilogue - the explicit store drain cost
  // 4 % and the neighbours did not wait any less.)
  auto publish = [&](int tile, unsigned pub_need) {
    if (tid == 0) {
      if (pub_need && !(ABL & 4)) {   // (an earlier unit of the same layer on this tile: it never waits for this one)
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(flags + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - pub_need) < 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > spin_limit) { raise_err(); break; }
        }
      }
      __hip_atomic_fetch_add(flags + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  // ---- first ticket
  if (tid == 0) ctl[0] = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned q = __builtin_amdgcn_readfirstlane(ctl[0]);
  if (q >= nwork) return;
  int item = (int)(q / (unsigned)ntiles), tile = (int)(q - (unsigned)item * (unsigned)ntiles);
  UnitP up = load_unit(ca.items, item);
  int n, y0, x0;
  setup_tile(tile, n, y0, x0);
  int par = 0;
  write_epi(par, up);
  bool have0 = false;   // this unit's first chunk is already in LDS (prefetched during the previous unit's last chunk)
  int pend = -1;        // tile whose counter add is pending (its stores may still be in flight)
  unsigned pend_need = 0;
  int buf = 0;

  struct Frags { uint4 wf[3]; uint4 af[MB + 2]; };

  while (true) {
    const UnitP cur = up;
    const int nchunks = cur.nch;
    if (!have0) {
      // blocking start of a unit: publish what is pending (never block while holding back a finished tile), wait for the
      // unit's first dependency, bring the first chunk in
      if (pend >= 0) {
        dma_wait();
        __syncthreads();
        publish(pend, pend_need);
        pend = -1;
      }
      if (ABL & 8) ++st_block;
      st_mid = false;
      if (wave == 0) poll_block(nb_index(tile), cur.need0);
      __syncthreads();
      buf = 0;
      prefetch_begin(cur, 0, 0);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) dma_op(i);
      dma_wait();
      __syncthreads();
    }
    // the workgroup's next ticket is taken as LATE as possible - three chunks before the end of this unit, one chunk before it
    // is needed: a ticket held while this unit computes (or waits) is a unit nobody else may start, and units that run out of
    // order eat the margin between a unit and the ones it waits for
    unsigned tk = 0;

    f32x16 acc[MB];
    {
      float bias_v[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 b4 = *reinterpret_cast<const float4*>(epi_lds + par * 64 + 16 * (qd >> 1) + 8 * lh + 4 * (qd & 1));
        bias_v[4 * qd] = b4.x; bias_v[4 * qd + 1] = b4.y; bias_v[4 * qd + 2] = b4.z; bias_v[4 * qd + 3] = b4.w;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] = bias_v[i];
    }
    const int cur_n = n, cur_y0 = y0, cur_x0 = x0, cur_tile = tile;
    const int xo = cur_x0 + lr;
    unsigned q_next = nwork; bool next_ready = false;
    int tile_next = 0;

    for (int c = 0; c < nchunks; ++c) {
      pf_on = false;
      if (c + 1 < nchunks) {
        prefetch_begin(cur, c + 1, buf ^ 1);
      } else if (next_ready) {
        setup_tile(tile_next, n, y0, x0);
        prefetch_begin(up, 0, buf ^ 1);   // `up` already holds the next unit's parameters
      }
      // polls that must be resolved by the barrier at the end of this chunk (wave 0): the flag loads are issued here, two
      // chunks ahead of the DMA they guard, and evaluated after this chunk's MFMAs
      if (c + 3 == nchunks && tid == 0) tk = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool due_new = c + 2 == cur.newest, due_next = c + 2 == nchunks;
      unsigned fv = 0, need = 0; int nbi = -1; unsigned qn = nwork;
      if (wave == 0 && (due_new || due_next)) {
        if (due_next) {
          qn = __builtin_amdgcn_readfirstlane(tk);
          if (qn < nwork) {
            const int itn = (int)(qn / (unsigned)ntiles), tn = (int)(qn - (unsigned)itn * (unsigned)ntiles);
            CItem nx = (CItem)(ca.items + itn);
            need = nx->newest == 0 ? nx->need_new : nx->need_old;
            nbi = nb_index(tn);
          }
        } else {
          need = cur.need_new; nbi = nb_index(cur_tile);
        }
        if (nbi >= 0) fv = __hip_atomic_load(flags + nbi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const char* tb = smem + buf * TILE_BYTES;
      const char* wb = smem + 2 * TILE_BYTES + buf * W_BYTES + lane * 16;
      // ONE fragment set, reloaded in place (conv_mfma.hip, ROLL): same MFMA order, same DMA slots
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
      if (c + 1 < nchunks) {
        dma_wait();       // next chunk has landed; this wave's stores of the previous unit have drained
        if (c == 0 && pend >= 0) {
          // the previous unit's counter add, as early as its stores allow - and BEFORE this unit may block below (a blocked
          // workgroup never holds back a finished tile: its neighbours may be waiting for exactly that tile)
          __syncthreads();
          publish(pend, pend_need);
          pend = -1;
        }
        if (wave == 0 && (due_new || due_next)) {
          const bool ok = (ABL & 4) ? true : satisfied(fv, nbi, need);
          if (due_new) {
            st_mid = true;
            if (!ok) poll_block(nbi, need);   // the previous layer is not through on the neighbourhood yet: wait here
          } else if (lane == 0) {
            ctl[0] = qn; ctl[1] = (qn < nwork && ok) ? 1u : 0u;
          }
        }
        __syncthreads();  // every wave is done reading this buffer (and has passed its vmcnt(0))
        buf ^= 1;
        if (due_next) {
          q_next = __builtin_amdgcn_readfirstlane(ctl[0]);
          next_ready = __builtin_amdgcn_readfirstlane(ctl[1]) != 0;
          if (q_next < nwork) {
            // the next unit's parameters come into registers here: nothing is in flight behind this barrier, so the waits
            // of these loads cost nothing (and `cur` keeps this unit's copy)
            const int item_next = (int)(q_next / (unsigned)ntiles);
            tile_next = (int)(q_next - (unsigned)item_next * (unsigned)ntiles);
            up = load_unit(ca.items, item_next);
            write_epi(par ^ 1, up);
          }
        }
      }
    }

    if (next_ready) {   // hand the LDS buffers to the next unit before the epilogue (conv_mfma.hip)
      dma_wait();
      __syncthreads();
      buf ^= 1;
    }

    // ---------------- epilogue: conv_mfma.hip's EK_PLAIN arithmetic; write-through stores, L1-bypassing residual loads
    {
      constexpr int HB = 16;
      int lhe = lh;
      asm volatile("" : "+v"(lhe));
      const float alpha = cur.alpha, gamma = cur.gamma;
      float slope_v[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 s4 = *reinterpret_cast<const float4*>(epi_lds + par * 64 + 32 + 16 * (qd >> 1) + 8 * lhe + 4 * (qd & 1));
        slope_v[4 * qd] = s4.x; slope_v[4 * qd + 1] = s4.y; slope_v[4 * qd + 2] = s4.z; slope_v[4 * qd + 3] = s4.w;
      }
      // wave-uniform base of this wave's rows inside each plane; per-lane byte offset of (row mb, pixel lr, half lh)
      const size_t wrec0 = (((size_t)cur_n * ca.H + cur_y0 + wave * MB) * ca.W + cur_x0) * REC;
      const uint32_t voff0 = (uint32_t)lr * REC + (uint32_t)lhe * HB;
      const uint32_t row_b = (uint32_t)ca.W * REC;
      char* outb = cur.out + wrec0;
      const __amdgpu_buffer_rsrc_t ro0 = rsrc_of(outb), ro1 = rsrc_of(outb + cur.out_pb);
      const bool has1 = cur.res1 != nullptr, has2 = cur.res2 != nullptr;
      if (!has1 && !has2) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const bool ok = (cur_y0 + wave * MB + mb) < ca.H && xo < ca.W;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float t = acc[mb][i];
            v[i] = fmaxf(t, t * slope_v[i]) * alpha;
          }
          if (ok) {
            store8_sc1(ro0, voff0 + mb * row_b, v);
            store8_sc1(ro1, voff0 + mb * row_b, v + 8);
          }
        }
      } else {
        const char* r1b = has1 ? cur.res1 + wrec0 : outb;
        const char* r2b = has2 ? cur.res2 + wrec0 : outb;
        const __amdgpu_buffer_rsrc_t r10 = rsrc_of(r1b), r11 = rsrc_of(r1b + (has1 ? cur.r1_pb : 0));
        const __amdgpu_buffer_rsrc_t r20 = rsrc_of(r2b), r21 = rsrc_of(r2b + (has2 ? cur.r2_pb : 0));
        constexpr int RB = 2;
#pragma unroll
        for (int mb0 = 0; mb0 < MB; mb0 += RB) {
          u32x4v r1v[RB][2], r2v[RB][2];
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const int mb = mb0 + j;
            if (mb < MB) {
              const bool ok = (cur_y0 + wave * MB + mb) < ca.H && xo < ca.W;
              const u32x4v z = {0u, 0u, 0u, 0u};
              const uint32_t vo = voff0 + mb * row_b;
              r1v[j][0] = (has1 && ok) ? __builtin_amdgcn_raw_buffer_load_b128(r10, vo, 0, AUX_SC1) : z;
              r1v[j][1] = (has1 && ok) ? __builtin_amdgcn_raw_buffer_load_b128(r11, vo, 0, AUX_SC1) : z;
              r2v[j][0] = (has2 && ok) ? __builtin_amdgcn_raw_buffer_load_b128(r20, vo, 0, AUX_SC1) : z;
              r2v[j][1] = (has2 && ok) ? __builtin_amdgcn_raw_buffer_load_b128(r21, vo, 0, AUX_SC1) : z;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < RB; ++j) {
            const int mb = mb0 + j;
            if (mb < MB) {
              const bool ok = (cur_y0 + wave * MB + mb) < ca.H && xo < ca.W;
              float v[16], r1[16], r2[16];
#pragma unroll
              for (int hq = 0; hq < 2; ++hq) {
                load8<T>(reinterpret_cast<const char*>(&r1v[j][hq]), r1 + 8 * hq);
                load8<T>(reinterpret_cast<const char*>(&r2v[j][hq]), r2 + 8 * hq);
              }
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                float t = acc[mb][i];
                const float neg = t * slope_v[i];
                t = t >= 0.f ? t : neg;
                t = t * alpha + r1[i];
                v[i] = t * gamma + r2[i];
              }
              if (ok) {
                store8_sc1(ro0, voff0 + mb * row_b, v);
                store8_sc1(ro1, voff0 + mb * row_b, v + 8);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    pend = cur_tile; pend_need = cur.pub_need;

    if (q_next >= nwork) break;
    q = q_next; tile = tile_next; par ^= 1;
    have0 = next_ready;
    if (!have0) setup_tile(tile, n, y0, x0);
  }
  // the last unit of this workgroup
  dma_wait();
  __syncthreads();
  publish(pend, pend_need);
  if ((ABL & 8) && tid == 0) {
    __hip_atomic_fetch_add(head + 2, st_block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(head + 3, st_spin0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(flags + ntiles, st_spin1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MB, int ABL = 0>
static void launch_t(ss4k_ctx* ctx, const ChainArgs& a, hipStream_t st) {
  constexpr size_t lds = lds_bytes_chain<MB>();
  constexpr int per_cu = chain_wgs_per_cu<MB>();
  static_assert(per_cu * lds <= 160 * 1024, "workgroups per CU");
  const void* fn = reinterpret_cast<const void*>(&conv3x3_chain_kernel<MB, ABL>);
  if (ctx->lds_attr_set.insert(fn).second)
    SS4K_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  // Units are queued, so any grid is CORRECT.  The fastest grid keeps fewer units in flight than lie between a unit and the
  // nearest unit it waits for, (L, t) -> (L - 1, t + tiles_x + 1), i.e. ntiles - tiles_x - 1 tickets: with more workgroups every
  // unit starts before its predecessor layer has reached it and the chain of waits becomes the critical path (measured on one
  // 720p frame, 460 tiles, 512 slots: 438-460 workgroups 94-97 fps, 480: 88-91, 512: 80-90; DESIGN.md 4.5, profiles/NOTES_r01_r03.md 4.1d)
  int gx = std::max(1, std::min(per_cu * ctx->num_cu, ntiles - a.tiles_x - 1));
  if (a.grid > 0) gx = a.grid;
  SS4K_HIP(hipMemsetAsync(a.ctl, 0, conv_chain_ctl_bytes(ntiles), st));
  hipLaunchKernelGGL((conv3x3_chain_kernel<MB, ABL>), dim3(gx), dim3(64 * NW), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

}  // namespace chain

size_t conv_chain_ctl_bytes(int ntiles) { return ((size_t)(4 + ntiles + 1) * 4 + 15) & ~size_t(15); }   // + one statistics word (dev)

int conv_chain_tiles(int N, int H, int W, int rows_per_wave, int* tiles_x, int* tiles_y) {
  const int th = 4 * rows_per_wave;
  *tiles_x = (W + TW - 1) / TW; *tiles_y = (H + th - 1) / th;
  return N * *tiles_x * *tiles_y;
}

void launch_conv_chain(ss4k_ctx* ctx, const ChainArgs& a, int rows_per_wave, hipStream_t st) {
  SS4K_REQUIRE(a.items && a.nitems > 0 && a.ctl && a.N > 0 && a.H > 0 && a.W > 0, "conv chain: empty");
  SS4K_REQUIRE((double)a.N * a.H * a.W < 2147483648.0 && (double)a.W * 32.0 * 24.0 < 2147483648.0, "conv chain: a plane holds at most 2^31 pixels");
  SS4K_REQUIRE((double)a.nitems * a.N * a.tiles_x * a.tiles_y < 4.0e9, "conv chain: too many work units");
#ifdef SS4K_DEV
  if (a.abl && rows_per_wave == 3) {
    switch (a.abl) {
      case 4: chain::launch_t<3, 4>(ctx, a, st); break;
      case 8: chain::launch_t<3, 8>(ctx, a, st); break;
      default: throw Error(SS4K_EINVAL, "chain ablation on 12-row tiles: 4 or 8");
    }
    return;
  }
  if (a.abl) {
    SS4K_REQUIRE(rows_per_wave == 4, "chain ablations are built for 16- and 12-row tiles");
    switch (a.abl) {
      case 1: chain::launch_t<4, 1>(ctx, a, st); break;
      case 2: chain::launch_t<4, 2>(ctx, a, st); break;
      case 3: chain::launch_t<4, 3>(ctx, a, st); break;
      case 4: chain::launch_t<4, 4>(ctx, a, st); break;
      case 7: chain::launch_t<4, 7>(ctx, a, st); break;
      case 8: chain::launch_t<4, 8>(ctx, a, st); break;
      default: throw Error(SS4K_EINVAL, "chain ablation: 1, 2, 3, 4, 7 or 8 (statistics)");
    }
    return;
  }
#endif
  SS4K_REQUIRE(a.abl == 0, "chain ablations live in libss4k_hip_dev.so only");
  if (rows_per_wave == 5) chain::launch_t<5>(ctx, a, st);
  else if (rows_per_wave == 3) chain::launch_t<3>(ctx, a, st);
  else chain::launch_t<4>(ctx, a, st);
}

}  // namespace ss4k
