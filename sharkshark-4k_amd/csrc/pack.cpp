// Host-side repacking of PyTorch OIHW conv weights into the MFMA fragment order the conv kernel
// DMAs straight into LDS (see conv_mfma.hip).  Pure host code: no HIP calls.
//
// Packed order: [group][chunk of 16 cin][dx(3) x ks(KS: 1 fp16, 2 fp32)][dy(3)][nb][lane(64)][E]
//   lane l: MFMA row rho = l & 31 (-> output channel, permuted so that result lane (pixel, h) holds
//   channels 8h..8h+7 of the block's first 16-channel plane in accumulator elements 0-7 and of its
//   second plane in elements 8-15), k half hk = l >> 5; element e: channel (2*ks + hk) * E + e of the chunk.
#include "common.h"
#include <cmath>

namespace ss4k {

static inline uint16_t f32_to_f16_bits(float f) {
  // round-to-nearest-even, handles subnormals/inf/nan
  uint32_t x; std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp;
    uint32_t h = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1))) ++h;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
  const uint32_t rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
  return (uint16_t)(sign | h);
}

int virt_to_real_cout(const PackSpec& s, int v) {
  if (v >= s.cout_real) return -1;
  if (s.ps2) {
    // PixelShuffle(2): real channel c*4 + dy*2 + dx lands at output channel c of pixel (2y+dy, 2x+dx).  Virtual
    // order (conv_mfma.hip, EK_PS2): 32-cout block b = (dy, 16 output channels oc0..oc0+15); its first
    // 16-channel plane is sub-pixel dx = 0, its second dx = 1 - so one lane holds the same 8 channels of two
    // horizontally adjacent output pixels and a wave's stores cover whole runs of output records
    const int cp = s.cout_real / 4, cpb = cp / 16;
    const int b = v / 32, dx = (v >> 4) & 1, j = v & 15;
    const int dy = b / cpb, c = 16 * (b - dy * cpb) + j;
    return c * 4 + dy * 2 + dx;
  }
  return v;
}

PackedConv pack_conv3x3(const PackSpec& s, const float* w, const float* bias, const float* prelu) {
  const int E = s.dtype == SS4K_F16 ? 8 : 4, CW = 16, KS = CW / (2 * E);  // k-steps per 16-channel chunk
  const int nchunks = s.nchunks0 + s.nchunks1;
  SS4K_REQUIRE((int)s.cin_map.size() == nchunks * CW, "pack_conv3x3: cin_map size");
  PackedConv p;
  p.nb = (s.cout_real <= 32 || s.force_nb1) ? 1 : 2;
  const int gw = p.nb * 32;
  const int padw = s.cout_real <= 32 ? 32 : 64;   // the tensor's planes come in 32- / 64-cout blocks whatever the group width
  p.cout_pad = (s.cout_real + padw - 1) / padw * padw;
  p.groups = p.cout_pad / gw;
  const size_t esz = s.dtype == SS4K_F16 ? 2 : 4;
  const size_t nelem = (size_t)p.groups * nchunks * 9 * KS * p.nb * 64 * E;
  p.w.assign(nelem * esz, 0);
  p.bias.assign(p.cout_pad, 0.f);
  p.prelu.assign(p.cout_pad, 0.f);
  for (int v = 0; v < p.cout_pad; ++v) {
    const int co = virt_to_real_cout(s, v);
    if (co >= 0) { p.bias[v] = bias ? bias[co] : 0.f; p.prelu[v] = prelu ? prelu[co] : 0.f; }
  }
  size_t idx = 0;
  for (int g = 0; g < p.groups; ++g)
    for (int c = 0; c < nchunks; ++c)
      for (int dx = 0; dx < 3; ++dx)
        for (int ks = 0; ks < KS; ++ks)
          for (int dy = 0; dy < 3; ++dy)
            for (int nb = 0; nb < p.nb; ++nb)
              for (int lane = 0; lane < 64; ++lane) {
                const int rho = lane & 31, hk = lane >> 5;
                const int v = (g * p.nb + nb) * 32 + 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + (rho & 3) + 4 * ((rho >> 3) & 1);
                const int co = virt_to_real_cout(s, v);
                for (int e = 0; e < E; ++e, ++idx) {
                  const int ci = s.cin_map[(size_t)c * CW + (2 * ks + hk) * E + e];
                  float val = 0.f;
                  if (ci >= 0 && co >= 0) val = w[((size_t)co * s.cin_total + ci) * 9 + dy * 3 + dx];
                  if (esz == 2) { const uint16_t h = f32_to_f16_bits(val); std::memcpy(&p.w[idx * 2], &h, 2); }
                  else std::memcpy(&p.w[idx * 4], &val, 4);
                }
              }
  return p;
}

// conv_rs.hip: every wave holds its slice of the layer's weights in registers as v_mfma_f32_16x16x32_f16
// A fragments.  Order [group][cout group cg][32-channel chunk c][tap dy*3+dx][cb][lane][8 fp16]:
//   lane l: cout row m = l & 15 of block cb, k-group kq = l >> 4; element e: channel 8*kq + e of the chunk
//   (= plane 2c + (kq >> 1), channel 8*(kq & 1) + e of that plane).  Row m of block cb of cout group cg is
//   virtual cout  g*COUT_WG + cg*cb_count*16 + (cb_count == 2 ? 8*(m >> 2) + 4*cb + (m & 3) : m)  so that
//   result lane (pixel, q) holds cb_count*4 CONSECUTIVE channels (16- or 8-byte stores).
std::vector<uint8_t> pack_conv3x3_rs(const PackSpec& s, const float* w, int cout_pad, int nch, int cbn, int CG) {
  SS4K_REQUIRE(s.dtype == SS4K_F16, "pack_conv3x3_rs: fp16 only");
  const int nplanes = s.nchunks0 + s.nchunks1;
  SS4K_REQUIRE((int)s.cin_map.size() == nplanes * 16 && nplanes <= 2 * nch, "pack_conv3x3_rs: cin_map size");
  const int COUT_WG = CG * cbn * 16;
  SS4K_REQUIRE(cout_pad % COUT_WG == 0, "pack_conv3x3_rs: cout_pad");
  const int groups = cout_pad / COUT_WG;
  std::vector<uint8_t> out((size_t)groups * CG * nch * 9 * cbn * 64 * 8 * 2, 0);
  size_t idx = 0;
  for (int g = 0; g < groups; ++g)
    for (int cg = 0; cg < CG; ++cg)
      for (int c = 0; c < nch; ++c)
        for (int t = 0; t < 9; ++t)
          for (int cb = 0; cb < cbn; ++cb)
            for (int lane = 0; lane < 64; ++lane) {
              const int m = lane & 15, kq = lane >> 4;
              const int vl = cbn == 2 ? 8 * (m >> 2) + 4 * cb + (m & 3) : m;
              const int co = virt_to_real_cout(s, g * COUT_WG + cg * cbn * 16 + vl);
              for (int e = 0; e < 8; ++e, ++idx) {
                const int plane = 2 * c + (kq >> 1), ch = 8 * (kq & 1) + e;
                const int ci = plane < nplanes ? s.cin_map[(size_t)plane * 16 + ch] : -1;
                float val = 0.f;
                if (ci >= 0 && co >= 0) val = w[((size_t)co * s.cin_total + ci) * 9 + t];
                const uint16_t h = f32_to_f16_bits(val);
                std::memcpy(&out[idx * 2], &h, 2);
              }
            }
  return out;
}

// conv_w16.hip: one 64-cout group of a layer as v_mfma_f32_16x16x32_f16 A fragments, [group][pair of K-chunks q][phase][dy][16-cout
// block mbk][lane][8 fp16].  lane l: MFMA row m = l & 15 of block mbk, k-group kq = l >> 4; element e: channel 8 (kq & 1) + e of
//   phase 0: plane 2q,            tap column dx = kq >> 1       (taps dx 0 | dx 1 of one plane)
//   phase 1: plane 2q + (kq >> 1), tap column dx = 2            (tap dx 2 of both planes)
//   phase 2: plane 2q + 1,        tap column dx = kq >> 1
// Row m = 4 rg + i of block mbk = 2 j + e2 is virtual cout 64 g + 32 j + 16 (rg >> 1) + 8 (rg & 1) + 4 e2 + i: result lane (pixel, rg)
// then holds channels 8 (rg & 1) .. + 7 of plane 2 j + (rg >> 1) in blocks 2 j and 2 j + 1 (one 16-byte store).
std::vector<uint8_t> pack_conv3x3_w16(const PackSpec& s, const float* w, int cout_pad) {
  SS4K_REQUIRE(s.dtype == SS4K_F16, "pack_conv3x3_w16: fp16 only");
  const int nplanes = s.nchunks0 + s.nchunks1;
  SS4K_REQUIRE((int)s.cin_map.size() == nplanes * 16 && nplanes % 2 == 0 && cout_pad % 64 == 0, "pack_conv3x3_w16: shape");
  const int groups = cout_pad / 64, np = nplanes / 2;
  std::vector<uint8_t> out((size_t)groups * np * 3 * 3 * 4 * 64 * 8 * 2, 0);
  size_t idx = 0;
  for (int g = 0; g < groups; ++g)
    for (int q = 0; q < np; ++q)
      for (int ph = 0; ph < 3; ++ph)
        for (int dy = 0; dy < 3; ++dy)
          for (int mbk = 0; mbk < 4; ++mbk)
            for (int lane = 0; lane < 64; ++lane) {
              const int m = lane & 15, kq = lane >> 4, rg = m >> 2, i = m & 3;
              const int co = virt_to_real_cout(s, g * 64 + 32 * (mbk >> 1) + 16 * (rg >> 1) + 8 * (rg & 1) + 4 * (mbk & 1) + i);
              const int plane = ph == 0 ? 2 * q : ph == 1 ? 2 * q + (kq >> 1) : 2 * q + 1;
              const int dx = ph == 1 ? 2 : (kq >> 1);
              for (int e = 0; e < 8; ++e, ++idx) {
                const int ci = s.cin_map[(size_t)plane * 16 + 8 * (kq & 1) + e];
                float val = 0.f;
                if (ci >= 0 && co >= 0) val = w[((size_t)co * s.cin_total + ci) * 9 + dy * 3 + dx];
                const uint16_t h = f32_to_f16_bits(val);
                std::memcpy(&out[idx * 2], &h, 2);
              }
            }
  return out;
}

// conv_w16n.hip: a layer with at most 16 output channels as ONE 16-row MFMA block, [pair of K-chunks q][phase][dy][lane][8 fp16]; phases and
// k-groups as pack_conv3x3_w16; row m is output channel m
std::vector<uint8_t> pack_conv3x3_w16n(const PackSpec& s, const float* w) {
  SS4K_REQUIRE(s.dtype == SS4K_F16 && s.cout_real <= 16 && !s.ps2, "pack_conv3x3_w16n: fp16, at most 16 output channels");
  const int nplanes = s.nchunks0 + s.nchunks1;
  SS4K_REQUIRE((int)s.cin_map.size() == nplanes * 16 && nplanes % 2 == 0, "pack_conv3x3_w16n: shape");
  std::vector<uint8_t> out((size_t)(nplanes / 2) * 3 * 3 * 64 * 8 * 2, 0);
  size_t idx = 0;
  for (int q = 0; q < nplanes / 2; ++q)
    for (int ph = 0; ph < 3; ++ph)
      for (int dy = 0; dy < 3; ++dy)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, kq = lane >> 4;
          const int plane = ph == 0 ? 2 * q : ph == 1 ? 2 * q + (kq >> 1) : 2 * q + 1;
          const int dx = ph == 1 ? 2 : (kq >> 1);
          for (int e = 0; e < 8; ++e, ++idx) {
            const int ci = s.cin_map[(size_t)plane * 16 + 8 * (kq & 1) + e];
            float val = 0.f;
            if (ci >= 0 && m < s.cout_real) val = w[((size_t)m * s.cin_total + ci) * 9 + dy * 3 + dx];
            const uint16_t h = f32_to_f16_bits(val);
            std::memcpy(&out[idx * 2], &h, 2);
          }
        }
  return out;
}

#ifdef SS4K_DEV
// conv_d16.hip: a dense-block layer pair (conv_k: K1 planes -> 32 couts; conv_{k+1}: the same K1 planes + x_k's two -> 32 couts) as
// v_mfma_f32_16x16x32_f16 A fragments.  K1 / 2 chunk pairs of three 12 KB phases [dy][conv_k b0, conv_{k+1} b0, conv_k b1, conv_{k+1} b1][lane][8],
// then conv_{k+1}'s x_k chunk pair as three 6 KB phases [dy][b0, b1][lane][8].  Phases and k-groups as pack_conv3x3_w16; row m of block b is
// cout 16 b + m (result lane (pixel, row group rg) holds channels 4 rg .. 4 rg + 3 of plane b).
std::vector<uint8_t> pack_dense_d16(const PackSpec& sa, const float* wa, const PackSpec& sb, const float* wb) {
  SS4K_REQUIRE(sa.dtype == SS4K_F16 && sb.dtype == SS4K_F16, "pack_dense_d16: fp16 only");
  const int k1 = sa.nchunks0 + sa.nchunks1, k2 = sb.nchunks0 + sb.nchunks1;
  SS4K_REQUIRE(k1 % 2 == 0 && k2 == k1 + 2 && (int)sa.cin_map.size() == k1 * 16 && (int)sb.cin_map.size() == k2 * 16, "pack_dense_d16: shape");
  std::vector<uint8_t> out((size_t)(k1 / 2) * 3 * 12288 + 3 * 6144, 0);
  size_t idx = 0;
  auto put = [&](const PackSpec& s, const float* w, int blk, int q, int ph, int dy, int lane) {
    const int m = lane & 15, kq = lane >> 4;
    const int co = virt_to_real_cout(s, 16 * blk + m);
    const int plane = ph == 0 ? 2 * q : ph == 1 ? 2 * q + (kq >> 1) : 2 * q + 1;
    const int dx = ph == 1 ? 2 : (kq >> 1);
    for (int e = 0; e < 8; ++e, ++idx) {
      const int ci = s.cin_map[(size_t)plane * 16 + 8 * (kq & 1) + e];
      float val = 0.f;
      if (ci >= 0 && co >= 0) val = w[((size_t)co * s.cin_total + ci) * 9 + dy * 3 + dx];
      const uint16_t h = f32_to_f16_bits(val);
      std::memcpy(&out[idx * 2], &h, 2);
    }
  };
  for (int q = 0; q < k1 / 2; ++q)
    for (int ph = 0; ph < 3; ++ph)
      for (int dy = 0; dy < 3; ++dy)
        for (int t = 0; t < 4; ++t)   // table entry t: layer t & 1 (0 = conv_k), block t >> 1
          for (int lane = 0; lane < 64; ++lane) put((t & 1) ? sb : sa, (t & 1) ? wb : wa, t >> 1, q, ph, dy, lane);
  for (int ph = 0; ph < 3; ++ph)
    for (int dy = 0; dy < 3; ++dy)
      for (int blk = 0; blk < 2; ++blk)
        for (int lane = 0; lane < 64; ++lane) put(sb, wb, blk, k1 / 2, ph, dy, lane);
  SS4K_REQUIRE(idx * 2 == out.size(), "pack_dense_d16: size");
  return out;
}
#endif  // SS4K_DEV

}  // namespace ss4k
