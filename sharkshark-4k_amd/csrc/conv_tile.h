// Building blocks shared by the 3x3 conv kernels of conv_mfma.hip (one launch per layer) and conv_chain.hip (the RRDB body as one
// persistent launch): tile geometry, LDS-DMA, record loads / stores, the MFMA wrapper.  Device code, gfx950 only.
#pragma once
#include "common.h"

namespace ss4k {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TW = 32, IN_W = TW + 2;
constexpr uint32_t OOB = 0xFFFFFFFFu;
// Tile geometry: NW waves x MB output rows per wave x 32 pixels.  Production fp16 builds are
// <NB=1,MB=4,NW=4> (32-cout layers, 58 KB LDS) and <NB=2,MB=4,NW=4> (64-cout layers, 76 KB LDS): a 16x32
// pixel tile at TWO workgroups per CU, so one workgroup's epilogue / barrier / DMA wait is covered by the
// other's MFMAs.  The other shapes (MB = 2: 8-row tiles at three workgroups per CU; NW = 8: 32-row
// tiles, one workgroup per CU) are compiled only into the dev library (SS4K_DEV, ss4k_bench_conv).
template <typename T> struct Tr;
// E elements per 16-byte slot; a 16-channel record is SPR slots; one MFMA k-step eats two of them
// (one per half-wave), so a K-chunk is KS k-steps per tap column
template <> struct Tr<__half> { static constexpr int E = 8, KS = 1; };
template <> struct Tr<float> { static constexpr int E = 4, KS = 2; };
constexpr int CW = 16;  // channels per plane = per K-chunk, both dtypes

template <typename T, int MB, int NW> struct Geo {
  static constexpr int KS = Tr<T>::KS, SPR = 2 * KS, REC = 16 * SPR;
  static constexpr int TH = NW * MB, IN_H = TH + 2, IN_PIX = IN_W * IN_H;
  static constexpr int TILE_SLOTS = IN_PIX * SPR;          // 16-byte LDS slots per halo tile
  static constexpr int TILE_DMA = (TILE_SLOTS + 63) / 64;  // wave-level DMA instructions per tile
  static constexpr int DMA_PER_WAVE = (TILE_DMA + NW - 1) / NW;
};
template <typename T, int NB, int MB, int NW> constexpr size_t lds_bytes() {
  return (size_t)(2 * Geo<T, MB, NW>::TILE_SLOTS + 2 * 9 * Tr<T>::KS * NB * 64) * 16 + NB * 32 * 2 * 4;
}
// workgroups of one build that fit a CU (160 KB LDS, at most 3 so that 4-wave builds keep >= 168 VGPRs)
template <typename T, int NB, int MB, int NW> constexpr int wgs_per_cu() {
  return NW == 8 ? 1 : (lds_bytes<T, NB, MB, NW>() <= 53 * 1024 ? 3 : lds_bytes<T, NB, MB, NW>() <= 80 * 1024 ? 2 : 1);
}

// One wave-level LDS-DMA: every lane moves 16 bytes from ITS global address to LDS byte
// lds_addr + 16*lane.  Written as inline asm so hipcc does not count it: the compiler would
// otherwise put s_waitcnt vmcnt(0) in front of the first ds_read that follows (it cannot prove the
// DMA target and the buffer being read are different halves of the one LDS array), which serialises
// the prefetch with the MFMAs.  Completion is awaited explicitly (dma_wait) before the barrier that
// hands the buffer to the readers.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_addr_wave_uniform) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr_wave_uniform)
               : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T> __device__ __forceinline__ void load16(const char* p, float* v);
template <> __device__ __forceinline__ void load16<__half>(const char* p, float* v) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 16);
  const __half* ha = reinterpret_cast<const __half*>(&a);
  const __half* hb = reinterpret_cast<const __half*>(&b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = __half2float(ha[i]); v[8 + i] = __half2float(hb[i]); }
}
template <> __device__ __forceinline__ void load16<float>(const char* p, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(p + 16 * q);
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
  }
}
template <typename T> __device__ __forceinline__ void store16(char* p, const float* v);
template <> __device__ __forceinline__ void store16<__half>(char* p, const float* v) {
  uint4 a, b;
  __half* ha = reinterpret_cast<__half*>(&a);
  __half* hb = reinterpret_cast<__half*>(&b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = __float2half(v[i]); hb[i] = __float2half(v[8 + i]); }
  *reinterpret_cast<uint4*>(p) = a;
  *reinterpret_cast<uint4*>(p + 16) = b;
}
template <> __device__ __forceinline__ void store16<float>(char* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(p + 16 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// one lane's 8-channel half record (16 bytes fp16, 32 bytes fp32)
template <typename T> __device__ __forceinline__ void load8(const char* p, float* v);
template <> __device__ __forceinline__ void load8<__half>(const char* p, float* v) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  const __half* ha = reinterpret_cast<const __half*>(&a);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __half2float(ha[i]);
}
template <> __device__ __forceinline__ void load8<float>(const char* p, float* v) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 a = *reinterpret_cast<const float4*>(p + 16 * q);
    v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
  }
}
// Output records are streamed with non-temporal stores: the next layer reads them back only after
// this launch, so keeping them in L2 just evicts the halo rows and weights this launch re-reads (+1.8 %).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void store8(char* p, const float* v);
template <> __device__ __forceinline__ void store8<__half>(char* p, const float* v) {
  uint4 a;
  __half* ha = reinterpret_cast<__half*>(&a);
#pragma unroll
  for (int i = 0; i < 8; ++i) ha[i] = __float2half(v[i]);
  __builtin_nontemporal_store(*reinterpret_cast<u32x4*>(&a), reinterpret_cast<u32x4*>(p));
}
template <> __device__ __forceinline__ void store8<float>(char* p, const float* v) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const float4 f = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&f), reinterpret_cast<u32x4*>(p + 16 * q));
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a full workgroup fence: it also drains the vector-memory counter
// (s_waitcnt vmcnt(0)), i.e. waits for every global load still in flight and every store - which serialises a row prefetch
// that is meant to stay in flight across several barriers.  Use where waves exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename T>
__device__ __forceinline__ f32x16 mma(const uint4& w, const uint4& x, f32x16 acc) {
  if constexpr (sizeof(T) == 2) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&w),
                                                  *reinterpret_cast<const f16x8*>(&x), acc, 0, 0, 0);
  } else {
    const float* wf = reinterpret_cast<const float*>(&w);
    const float* xf = reinterpret_cast<const float*>(&x);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], xf[j], acc, 0, 0, 0);
    return acc;
  }
}

}  // namespace ss4k
