// 3x3 convolution over NHWC activations as an im2col-free implicit GEMM on the CDNA4 matrix
// cores.  D[cout][pixel] += W[cout][k] * X[k][pixel], k = (tap, cin):
//   * A operand = weights, pre-packed on the host in exact MFMA fragment order (pack.cpp), with
//     the MFMA row -> cout map chosen so each lane ends up owning 16 CONTIGUOUS output channels
//     of one pixel (vector NHWC stores, fused bias/activation/residual epilogue in registers);
//   * B operand = activations: a (TH+2)x(TW+2) halo tile of one K-chunk (64 B of channels per
//     pixel) staged in LDS, XOR-swizzled so every ds_read_b128 is bank-conflict free; the nine
//     taps are nine shifted reads of the same tile (no im2col, no re-fetch from HBM);
//   * f16: v_mfma_f32_32x32x16_f16 (fp32 accumulate); f32: v_mfma_f32_32x32x2_f32 (exact fp32).
// One workgroup = 4 waves = 8x32 output pixels x (NB*32) output channels; waves split the rows.
// The dense-block concat of RRDBNet is two input segments (no copy); nearest-x2 upsampling,
// stride-2 subsampling, PixelShuffle and the NCHW fp32 hand-off are address modes, not passes.
#include "common.h"

namespace ss4k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TW = 32, TH = 8, MB = 2;
constexpr int IN_W = TW + 2, IN_H = TH + 2, IN_PIX = IN_W * IN_H, IN_SLOTS = IN_PIX * 4;
// LDS image of the halo tile: four planes (one per 16-byte channel group g) of IN_PIX slots.
// A wave's MFMA operand read touches 32 consecutive pixels of one plane = 512 contiguous bytes
// (bank-conflict free) and every tap/k-step is a compile-time offset from ONE per-lane base.
// The plane stride is padded to 32 B mod 128 B so the staging ds_write_b128 (lanes = 4 channel
// groups x 2 pixels per 8-lane group) is conflict free too.
constexpr int PLANE = IN_PIX + 6;
static_assert((PLANE * 16) % 128 == 32, "plane stride must be 32 mod 128 bytes");
constexpr int IN_LDS_SLOTS = PLANE * 4;
constexpr int NTHREADS = 256;
constexpr int NLOAD_IN = (IN_SLOTS + NTHREADS - 1) / NTHREADS;

template <typename T> struct Tr;
template <> struct Tr<__half> { static constexpr int E = 8, KC = 32; };
template <> struct Tr<float> { static constexpr int E = 4, KC = 16; };

int conv_kc(int dtype) { return dtype == SS4K_F16 ? 32 : 16; }

__device__ __forceinline__ int in_slot(int p, int g) { return g * PLANE + p; }

template <typename T>
__device__ __forceinline__ void load16(const T* p, float* v);
template <>
__device__ __forceinline__ void load16<__half>(const __half* p, float* v) {
  uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
  const __half* ha = reinterpret_cast<const __half*>(&a);
  const __half* hb = reinterpret_cast<const __half*>(&b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = __half2float(ha[i]); v[8 + i] = __half2float(hb[i]); }
}
template <>
__device__ __forceinline__ void load16<float>(const float* p, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float4 a = *reinterpret_cast<const float4*>(p + 4 * q);
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
  }
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const float* v);
template <>
__device__ __forceinline__ void store16<__half>(__half* p, const float* v) {
  uint4 a, b;
  __half* ha = reinterpret_cast<__half*>(&a);
  __half* hb = reinterpret_cast<__half*>(&b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = __float2half(v[i]); hb[i] = __float2half(v[8 + i]); }
  *reinterpret_cast<uint4*>(p) = a;
  *reinterpret_cast<uint4*>(p + 8) = b;
}
template <>
__device__ __forceinline__ void store16<float>(float* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

template <typename T, int NB>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[NB][MB], const uint4 (&wf)[NB], const uint4 (&af)[MB]);

template <int NB>
__device__ __forceinline__ void mma_step_f16(f32x16 (&acc)[NB][MB], const uint4 (&wf)[NB], const uint4 (&af)[MB]) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&wf[nb]),
                                                           *reinterpret_cast<const f16x8*>(&af[mb]),
                                                           acc[nb][mb], 0, 0, 0);
}
template <int NB>
__device__ __forceinline__ void mma_step_f32(f32x16 (&acc)[NB][MB], const uint4 (&wf)[NB], const uint4 (&af)[MB]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float w = reinterpret_cast<const float*>(&wf[nb])[j];
        const float x = reinterpret_cast<const float*>(&af[mb])[j];
        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, acc[nb][mb], 0, 0, 0);
      }
}

template <typename T, int NB>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_kernel(const ConvArgs a) {
  constexpr int E = Tr<T>::E, KC = Tr<T>::KC;
  constexpr int WSLOTS = 9 * 2 * NB * 64;
  constexpr int NLOAD_W = (WSLOTS + NTHREADS - 1) / NTHREADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds_in = reinterpret_cast<uint4*>(smem);
  uint4* lds_w = lds_in + IN_LDS_SLOTS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int grp = blockIdx.y;
  const int nchunks = a.nchunks0 + a.nchunks1;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  const uint4* wbase = reinterpret_cast<const uint4*>(a.wpk) + (size_t)grp * nchunks * WSLOTS;
  const int Hs = a.ups2 ? (a.H >> 1) : a.H, Ws = a.ups2 ? (a.W >> 1) : a.W;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, tyn = tile / a.tiles_x;
    const int ty = tyn % a.tiles_y, n = tyn / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // source pixel (linear index) of every halo-tile slot this thread stages, -1 = zero padding
    int pix[NLOAD_IN];
#pragma unroll
    for (int i = 0; i < NLOAD_IN; ++i) {
      const int s = tid + NTHREADS * i;
      const int p = s >> 2;
      const int py = p / IN_W, px = p - py * IN_W;
      const int iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = (s < IN_SLOTS) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int sy = a.ups2 ? (iy >> 1) : iy, sx = a.ups2 ? (ix >> 1) : ix;
      pix[i] = ok ? (n * Hs + sy) * Ws + sx : -1;
    }

    f32x16 acc[NB][MB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][mb][i] = 0.f;

    uint4 rin[NLOAD_IN], rw[NLOAD_W];
    auto stage_load = [&](int c) {
      const T* base; int cs, ch, nslots;
      if (c < a.nchunks0) {
        base = reinterpret_cast<const T*>(a.in0); cs = a.cs0; ch = a.co0 + c * KC;
        nslots = min(4, (a.nch0 - c * KC) / E);
      } else {
        const int c1 = c - a.nchunks0;
        base = reinterpret_cast<const T*>(a.in1); cs = a.cs1; ch = a.co1 + c1 * KC;
        nslots = min(4, (a.nch1 - c1 * KC) / E);
      }
#pragma unroll
      for (int i = 0; i < NLOAD_IN; ++i) {
        const int g = (tid + NTHREADS * i) & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pix[i] >= 0 && g < nslots)
          v = *reinterpret_cast<const uint4*>(base + (size_t)pix[i] * cs + ch + g * E);
        rin[i] = v;
      }
      const uint4* wsrc = wbase + (size_t)c * WSLOTS;
#pragma unroll
      for (int i = 0; i < NLOAD_W; ++i) {
        const int s = tid + NTHREADS * i;
        if (s < WSLOTS) rw[i] = wsrc[s];
      }
      return nslots;
    };
    auto stage_write = [&]() {
#pragma unroll
      for (int i = 0; i < NLOAD_IN; ++i) {
        const int s = tid + NTHREADS * i;
        if (s < IN_SLOTS) lds_in[in_slot(s >> 2, s & 3)] = rin[i];
      }
#pragma unroll
      for (int i = 0; i < NLOAD_W; ++i) {
        const int s = tid + NTHREADS * i;
        if (s < WSLOTS) lds_w[s] = rw[i];
      }
    };

    const uint4* lds_a = lds_in + lh * PLANE + (wave * MB) * IN_W + lr;
    int nslots_cur = stage_load(0);
    stage_write();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      int nslots_next = 0;
      if (c + 1 < nchunks) nslots_next = stage_load(c + 1);
      const int nks = nslots_cur >> 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (ks < nks) {
            uint4 wf[NB], af[MB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[nb] = lds_w[((tap * 2 + ks) * NB + nb) * 64 + lane];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              af[mb] = lds_a[(2 * ks) * PLANE + (mb + dy) * IN_W + dx];
            }
            if constexpr (sizeof(T) == 2) mma_step_f16<NB>(acc, wf, af);
            else mma_step_f32<NB>(acc, wf, af);
          }
        }
      }
      __syncthreads();
      if (c + 1 < nchunks) {
        stage_write();
        __syncthreads();
      }
      nslots_cur = nslots_next;
    }

    // ---------------- epilogue: bias, activation, residuals, layout-aware store ----------------
    const int x = x0 + lr;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int y = y0 + wave * MB + mb;
      if (y >= a.H || x >= a.W) continue;
      const size_t ipix = ((size_t)n * a.H + y) * a.W + x;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int vbase = (grp * NB + nb) * 32 + 16 * lh;
        if (vbase >= a.cout_pad) continue;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[nb][mb][i] + a.bias[vbase + i];
        if (a.act == ACT_LRELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = v[i] >= 0.f ? v[i] : v[i] * a.slope;
        } else if (a.act == ACT_PRELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = v[i] >= 0.f ? v[i] : v[i] * a.prelu[vbase + i];
        } else if (a.act == ACT_RELU6) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fminf(fmaxf(v[i], 0.f), 6.f);
        }
        // where this 16-channel group lands
        size_t opix = ipix; int oc = vbase; bool keep = true;
        if (a.epi == EPI_NHWC_SUB2) {
          keep = !((y | x) & 1);
          opix = ((size_t)n * ((a.H + 1) >> 1) + (y >> 1)) * ((a.W + 1) >> 1) + (x >> 1);
        } else if (a.epi == EPI_NHWC_PS2) {
          const int cp = a.cout_real >> 2, sub = vbase / cp;
          oc = vbase - sub * cp;
          opix = ((size_t)n * 2 * a.H + 2 * y + (sub >> 1)) * (2 * a.W) + 2 * x + (sub & 1);
          keep = vbase < a.cout_real;
        }
        if (!keep) continue;
        if (a.alpha != 1.f) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] *= a.alpha;
        }
        if (a.res1 && (!a.bsvd_resid || vbase == 0)) {
          float r[16];
          const bool nhwc_dst = a.epi <= EPI_NHWC_PS2;
          const size_t rp = nhwc_dst ? opix : ipix;
          load16<T>(reinterpret_cast<const T*>(a.res1) + rp * a.r1cs + a.r1co + oc, r);
          if (a.bsvd_resid) {
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = r[i] - v[i];
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] += r[i];
          }
        }
        if (a.res2) {
          float r[16];
          load16<T>(reinterpret_cast<const T*>(a.res2) + opix * a.r2cs + a.r2co + oc, r);
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = v[i] * a.gamma + r[i];
        }
        if (a.epi <= EPI_NHWC_PS2) {
          if (oc + 16 <= a.cout_alloc)
            store16<T>(reinterpret_cast<T*>(a.out) + opix * a.ocs + a.oco + oc, v);
        } else {  // EPI_NCHW_F32
          float* o = reinterpret_cast<float*>(a.out);
          const size_t plane = (size_t)a.H * a.W;
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (vbase + i < a.cout_real) o[((size_t)n * a.cout_real + vbase + i) * plane + (size_t)y * a.W + x] = v[i];
        }
      }
    }
  }
}

template <typename T, int NB>
static void launch_t(ss4k_ctx* ctx, const ConvArgs& a, int groups, hipStream_t st) {
  constexpr size_t lds = (size_t)(IN_LDS_SLOTS + 9 * 2 * NB * 64) * 16;
  const int ntiles = a.N * a.tiles_y * a.tiles_x;
  static bool attr_set = false;
  if (!attr_set) {
    SS4K_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<T, NB>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int gx = ntiles;
  const int cap = ctx->num_cu * 2 * 4;  // persistent loop beyond this many workgroups
  if (gx * groups > cap) gx = std::max(1, cap / groups);
  hipLaunchKernelGGL((conv3x3_kernel<T, NB>), dim3(gx, groups), dim3(NTHREADS), lds, st, a);
  SS4K_HIP(hipGetLastError());
}

void launch_conv3x3(ss4k_ctx* ctx, const ConvArgs& a0, int dtype, hipStream_t st) {
  ConvArgs a = a0;
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  const int nb = a.cout_pad <= 32 ? 1 : 2;
  const int groups = (a.cout_pad + nb * 32 - 1) / (nb * 32);
  SS4K_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0, "conv3x3: empty grid");
  SS4K_REQUIRE(!a.ups2 || ((a.H % 2 == 0) && (a.W % 2 == 0)), "conv3x3: ups2 needs even grid");
  ProfEvent pe{};
  if (ctx->prof) {
    if (!ctx->prof_pool.empty()) { pe = ctx->prof_pool.back(); ctx->prof_pool.pop_back(); }
    else { SS4K_HIP(hipEventCreate(&pe.a)); SS4K_HIP(hipEventCreate(&pe.b)); }
    SS4K_HIP(hipEventRecord(pe.a, st));
  }
  if (dtype == SS4K_F16) {
    if (nb == 1) launch_t<__half, 1>(ctx, a, groups, st); else launch_t<__half, 2>(ctx, a, groups, st);
  } else {
    if (nb == 1) launch_t<float, 1>(ctx, a, groups, st); else launch_t<float, 2>(ctx, a, groups, st);
  }
  if (ctx->prof) {
    SS4K_HIP(hipEventRecord(pe.b, st));
    pe.flops = a0.flops;
    ctx->prof_events.push_back(pe);
  }
}

}  // namespace ss4k
