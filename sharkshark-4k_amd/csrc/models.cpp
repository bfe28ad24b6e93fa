// Network executors: FSRCNN, RRDBNet, SRVGGNetCompact and BSVD(F=1) as sequences of launches of
// the kernels in conv_mfma.hip / fsrcnn.hip.  Weights arrive as the reference's state_dict
// flattened in key order (see sharkshark-4k_amd/weights.py) and are repacked once at creation.
#include "models.h"
#ifdef SS4K_DEV
#include "chain_plan.h"
#endif
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace ss4k {

static size_t esz(int dtype) { return dtype == SS4K_F16 ? 2 : 4; }
static int pad16(int c) { return (c + 15) / 16 * 16; }

// ------------------------------------------------------------------------------------------
// parameter bookkeeping (must mirror weights.py: *_keys order)
// ------------------------------------------------------------------------------------------
struct ParamCursor {
  const float* base; size_t n, pos = 0;
  const float* take(size_t k) {
    SS4K_REQUIRE(pos + k <= n, "weight blob shorter than the model's state_dict");
    const float* p = base + pos; pos += k; return p;
  }
};

static std::vector<std::pair<int, int>> bsvd_denblock_shapes(const int* chns, int in_ch, int out_ch, int interm) {
  const int c0 = chns[0], c1 = chns[1], c2 = chns[2];
  // (cout, cin) in weights.py::_bsvd_denblock_convs order
  return {{interm, in_ch}, {c0, interm}, {c1, c0}, {c1, c1}, {c1, c1}, {c2, c1}, {c2, c2}, {c2, c2},
          {c2, c2}, {c2, c2}, {c1 * 4, c2}, {c1, c1}, {c1, c1}, {c0 * 4, c1}, {c0, c0}, {out_ch, c0}};
}

size_t model_param_count(const ss4k_model_desc& d) {
  switch (d.kind) {
    case SS4K_FSRCNN:
      return 56 * 25 + 56 + 56 + 12 * 56 + 12 + 12 + 4 * (12 * 12 * 9 + 12 + 12) + 56 * 12 + 56 + 56 + 56 * 81 + 1;
    case SS4K_RRDBNET: {
      const int nf = d.num_feat, g = d.num_grow_ch;
      const int cin0 = 3 * (d.scale == 2 ? 4 : d.scale == 1 ? 16 : 1);
      size_t n = (size_t)nf * cin0 * 9 + nf;
      size_t rdb = 0;
      for (int c = 0; c < 5; ++c) {
        const int ci = nf + c * g, co = c < 4 ? g : nf;
        rdb += (size_t)co * ci * 9 + co;
      }
      n += (size_t)d.num_block * 3 * rdb;
      n += 4 * ((size_t)nf * nf * 9 + nf);  // conv_body, conv_up1, conv_up2, conv_hr
      n += (size_t)3 * nf * 9 + 3;
      return n;
    }
    case SS4K_SRVGG: {
      const int nf = d.num_feat, r2 = d.scale * d.scale;
      size_t n = (size_t)nf * 3 * 9 + nf + nf;
      n += (size_t)d.num_block * ((size_t)nf * nf * 9 + nf + nf);
      n += (size_t)3 * r2 * nf * 9 + 3 * r2;
      return n;
    }
    case SS4K_BSVD: {
      size_t n = 0;
      for (int blk = 0; blk < 2; ++blk) {
        const int ci = blk == 0 ? 4 : d.bsvd_mid_ch, co = blk == 0 ? d.bsvd_mid_ch : 3;
        for (auto& s : bsvd_denblock_shapes(d.bsvd_chns, ci, co, d.bsvd_interm_ch))
          n += (size_t)s.first * s.second * 9 + s.first;
      }
      return n;
    }
  }
  return 0;
}

static void validate_desc(const ss4k_model_desc& d) {
  SS4K_REQUIRE(d.dtype == SS4K_F32 || d.dtype == SS4K_F16, "desc.dtype must be SS4K_F32 or SS4K_F16");
  SS4K_REQUIRE(d.bsvd_stream == 0 || (d.bsvd_stream == 1 && d.kind == SS4K_BSVD), "desc.bsvd_stream is 0 or 1 and only applies to BSVD");
  switch (d.kind) {
    case SS4K_FSRCNN:
      SS4K_REQUIRE(d.scale == 2 || d.scale == 4, "FSRCNN scale must be 2 or 4");
      break;
    case SS4K_RRDBNET:
      SS4K_REQUIRE(d.scale == 1 || d.scale == 2 || d.scale == 4, "RRDBNet scale must be 1, 2 or 4");
      SS4K_REQUIRE(d.num_feat > 0 && d.num_feat % 32 == 0 && d.num_grow_ch > 0 && d.num_grow_ch % 32 == 0 && d.num_block > 0,
                   "RRDBNet: num_feat and num_grow_ch must be multiples of 32");
      break;
    case SS4K_SRVGG:
      SS4K_REQUIRE(d.scale == 2 || d.scale == 4, "SRVGG upscale must be 2 or 4 (PixelShuffle tail)");
      SS4K_REQUIRE(d.num_feat > 0 && d.num_feat % 16 == 0 && d.num_block >= 0, "SRVGG: num_feat must be a multiple of 16");
      break;
    case SS4K_BSVD:
      SS4K_REQUIRE(d.bsvd_chns[0] % 32 == 0 && d.bsvd_chns[1] % 64 == 0 && d.bsvd_chns[2] % 64 == 0 && d.bsvd_mid_ch % 32 == 0 &&
                       d.bsvd_interm_ch > 0 && d.bsvd_interm_ch <= 256,
                   "BSVD: unsupported channel configuration");
      break;
    default:
      throw Error(SS4K_EINVAL, "unknown model kind");
  }
}

// ------------------------------------------------------------------------------------------
static void upload(DevBuf& b, const void* src, size_t bytes) {
  b.ensure(std::max<size_t>(bytes, 16));
  SS4K_HIP(hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice));
}

#ifdef SS4K_DEV
// bit of a layer shape in Model::rs_mask: 0-2 = 32 couts with 2/3/4 K-chunks (RDB conv1-3), 3 = conv4, 4 = 64->64, 5 = conv5
static int rs_shape_bit(int nch, int cout_pad) { return cout_pad == 32 ? nch - 2 : (nch == 2 ? 4 : 5); }
#endif

int Model::add_conv(ParamCursor& pc, int cout, int cin_total, PackSpec s, bool has_prelu_after, bool allow_rs, bool chainable) {
  s.dtype = desc.dtype; s.cout_real = cout; s.cin_total = cin_total;
  const float* w = pc.take((size_t)cout * cin_total * 9);
  const float* b = pc.take(cout);
  const float* a = has_prelu_after ? pc.take(cout) : nullptr;
  PackedConv p = pack_conv3x3(s, w, b, a);
  ConvLayer L;
  upload(L.w, p.w.data(), p.w.size());
  upload(L.bias, p.bias.data(), p.bias.size() * 4);
  if (a) upload(L.prelu, p.prelu.data(), p.prelu.size() * 4);
#ifdef SS4K_DEV   // conv_rs.hip (register-stationary weights) and conv_chain.hip (cross-layer chain): dev library only since round 5
  int nch, rows, cb, cg;
  if (allow_rs && use_rs && desc.dtype == SS4K_F16 && rs_config(s.nchunks0 + s.nchunks1, p.cout_pad, rs_wide, &nch, &rows, &cb, &cg) &&
      (rs_mask >> rs_shape_bit(nch, p.cout_pad)) & 1) {
    const std::vector<uint8_t> wr = pack_conv3x3_rs(s, w, p.cout_pad, nch, cb, cg);
    upload(L.wrs, wr.data(), wr.size());
    weight_bytes += wr.size();
    L.rs_wide = rs_wide;
  }
  if (chainable && chain_mode == 2 && desc.dtype == SS4K_F16 && p.nb == 2) {
    PackSpec s1 = s; s1.force_nb1 = 1;
    const PackedConv p1 = pack_conv3x3(s1, w, b, a);
    upload(L.wch, p1.w.data(), p1.w.size());
    weight_bytes += p1.w.size();
  }
#else
  (void)allow_rs; (void)chainable;
#endif
  if (use_w16 && use_wide && desc.dtype == SS4K_F16 && p.nb == 2 && p.cout_pad % 64 == 0 && (s.nchunks0 + s.nchunks1) % 2 == 0 && !s.ps2) {
    const std::vector<uint8_t> w6 = pack_conv3x3_w16(s, w, p.cout_pad);
    upload(L.w16, w6.data(), w6.size());
    weight_bytes += w6.size();
  }
  if (use_w16 && desc.dtype == SS4K_F16 && cout <= 4 && (s.nchunks0 + s.nchunks1) % 2 == 0 && !s.ps2) {   // conv_w16n.hip (a final 64 -> 3 layer)
    const std::vector<uint8_t> w6 = pack_conv3x3_w16n(s, w);
    upload(L.w16, w6.data(), w6.size());
    weight_bytes += w6.size();
  }
  L.has_prelu = a != nullptr;
  if (a) { L.prelu_le1 = true; for (int c = 0; c < cout; ++c) L.prelu_le1 = L.prelu_le1 && a[c] <= 1.f; }
  L.cout_real = cout; L.cout_pad = p.cout_pad; L.nchunks0 = s.nchunks0; L.nchunks1 = s.nchunks1;
  L.cin_real = 0;
  for (int c : s.cin_map) L.cin_real += c >= 0;
  weight_bytes += p.w.size() + p.bias.size() * 4;
#ifdef SS4K_DEV
  // dev experiment (SS4K_D16=1): a dense-block pair (this layer = conv_{k+1} of the previous one) once more in conv_d16.hip's order
  if (chainable && use_d16 && use_w16 && desc.dtype == SS4K_F16 && !layers.empty() && raw_w_prev && (int)layers.size() == raw_li_prev + 1 &&
      conv3x3_d16_eligible(layers.back().nchunks0 + layers.back().nchunks1, layers.back().cout_pad, s.nchunks0 + s.nchunks1, p.cout_pad) &&
      layers.back().nchunks0 == s.nchunks0 && !layers.back().w16p_is_second) {
    const std::vector<uint8_t> wp = pack_dense_d16(raw_s_prev, raw_w_prev, s, w);
    upload(layers.back().w16p, wp.data(), wp.size());
    weight_bytes += wp.size();
    L.w16p_is_second = true;
  }
  raw_w_prev = chainable ? w : nullptr; raw_s_prev = s; raw_li_prev = (int)layers.size();
#endif
  layers.push_back(std::move(L));
  return (int)layers.size() - 1;
}

// all of a tensor's channels, padded to whole planes
PackSpec Model::spec_plain(int cin_real, int ps2) const {
  PackSpec s{}; s.ps2 = ps2; s.nchunks0 = planes_for(cin_real); s.nchunks1 = 0;
  s.cin_map.assign((size_t)s.nchunks0 * cw(), -1);
  for (int j = 0; j < cin_real; ++j) s.cin_map[j] = j;
  return s;
}
// torch.cat((x, growth[:c1]), 1): segment 0 = x (c0 channels), segment 1 = the growth planes
PackSpec Model::spec_concat(int c0, int c1) const {
  PackSpec s{}; s.nchunks0 = planes_for(c0); s.nchunks1 = planes_for(c1);
  s.cin_map.assign((size_t)(s.nchunks0 + s.nchunks1) * cw(), -1);
  for (int j = 0; j < c0; ++j) s.cin_map[j] = j;
  for (int j = 0; j < c1; ++j) s.cin_map[(size_t)s.nchunks0 * cw() + j] = c0 + j;
  return s;
}
// BiBufferConv fed one frame: input channels [0, c/4) are zeros (bsvd/model.py:50-52,108,123):
// start at the first plane that holds a live channel and give the dead ones zero weights
PackSpec Model::spec_masked(int c) const {
  PackSpec s{}; const int first = (c / 4) / cw();
  s.nchunks0 = planes_for(c) - first; s.nchunks1 = 0;
  s.cin_map.assign((size_t)s.nchunks0 * cw(), -1);
  for (int j = 0; j < s.nchunks0 * cw(); ++j) {
    const int ch = first * cw() + j;
    if (ch >= c / 4 && ch < c) s.cin_map[j] = ch;
  }
  return s;
}

// BiBufferConv inside a stream: every input channel is live; the leading planes (channels < c/4,
// rounded up to whole planes) come from the time-shifted copy (segment 0), the rest from the tensor
PackSpec Model::spec_shifted(int c) const {
  PackSpec s{}; const int lead = shifted_planes(c);
  s.nchunks0 = lead; s.nchunks1 = planes_for(c) - lead;
  s.cin_map.assign((size_t)planes_for(c) * cw(), -1);
  for (int j = 0; j < c; ++j) s.cin_map[j] = j;
  return s;
}

void Model::build(const float* w, size_t n) {
  validate_desc(desc);
  // Routing choices that never change results are fields of the model description (include/ss4k.h, SS4K_MODEL_*).  Two of them
  // can also be set from the environment of a deployed service: SS4K_LANES (0 = measured per shape, 1 = one
  // launch chain, 2 = two) and SS4K_FS_EXACT=1.  The measurement-only switches exist in the dev library alone.
  const int fl = desc.flags;
#ifdef SS4K_DEV
  SS4K_REQUIRE((fl & ~(SS4K_MODEL_FLAGS_ALL | SS4K_DEV_MODEL_FLAGS_ALL)) == 0, "desc.flags: unknown SS4K_MODEL_* / SS4K_DEV_MODEL_* bit");
  SS4K_REQUIRE(!((fl & SS4K_DEV_MODEL_CHAIN) && (fl & SS4K_DEV_MODEL_CONV5_RS)), "desc.flags: the chain runs every layer on the 32-cout LDS-weights tile: CHAIN and CONV5_RS exclude each other");
  if (fl & SS4K_DEV_MODEL_CHAIN) chain_mode = 2;
  if (fl & SS4K_DEV_MODEL_CONV5_RS) { use_rs = true; conv5_mode = 1; }
#else
  SS4K_REQUIRE((fl & ~SS4K_MODEL_FLAGS_ALL) == 0, "desc.flags: unknown SS4K_MODEL_* bit (bits 8, 64, 128, 2048 and 16384 were retired with ABI 3)");
#endif
  SS4K_REQUIRE(!((fl & SS4K_MODEL_ONE_CHAIN) && (fl & SS4K_MODEL_TWO_CHAINS)), "desc.flags: ONE_CHAIN and TWO_CHAINS exclude each other");
  SS4K_REQUIRE(!((fl & SS4K_MODEL_TILE_ROWS_16) && (fl & SS4K_MODEL_TILE_ROWS_20)), "desc.flags: TILE_ROWS_16 and TILE_ROWS_20 exclude each other");
  if (fl & SS4K_MODEL_FS_EXACT) fs_exact = true;
  if (fl & SS4K_MODEL_ONE_CHAIN) lanes_mode = 1;
  if (fl & SS4K_MODEL_TWO_CHAINS) lanes_mode = 2;
  if (fl & SS4K_MODEL_TILE_ROWS_16) mb_override = 4;
  if (fl & SS4K_MODEL_TILE_ROWS_20) mb_override = 5;
  if (fl & SS4K_MODEL_NO_PAIR) use_pair = false;
  if (fl & SS4K_MODEL_NO_DENSE) dense_mode = 1;
  if (fl & SS4K_MODEL_HR_F32) hr_f32 = true;
  if (fl & SS4K_MODEL_NO_WIDE) use_wide = false;
  if (fl & SS4K_MODEL_NO_UPS_PRESUM) ups_presum = false;
  if (fl & SS4K_MODEL_NO_W16) use_w16 = false;
  if (!(fl & (SS4K_MODEL_ONE_CHAIN | SS4K_MODEL_TWO_CHAINS)))
    if (const char* e = std::getenv("SS4K_LANES")) lanes_mode = std::max(0, std::min(2, std::atoi(e)));
  if (const char* e = std::getenv("SS4K_FS_EXACT")) fs_exact = fs_exact || e[0] == '1';
#ifdef SS4K_DEV
  if (const char* e = std::getenv("SS4K_NO_FLIP")) flip_walk = !(e[0] == '1');  // A/B switch for the tile-walk direction
  if (const char* e = std::getenv("SS4K_SUBBATCH")) sub_batch = std::atoi(e);   // A/B switch: frames per pass through the network
  if (const char* e = std::getenv("SS4K_RS_MASK")) { rs_mask = std::atoi(e); use_rs = rs_mask != 0; }   // A/B switch: which layer shapes take conv_rs.hip
  if (const char* e = std::getenv("SS4K_RS_W8")) rs_wide = e[0] == '1';          // A/B switch: eight-wave variants of the 32-cout shapes
  if (const char* e = std::getenv("SS4K_MB")) mb_override = std::atoi(e);
  if (const char* e = std::getenv("SS4K_S3")) use_s3 = e[0] == '1';
  if (const char* e = std::getenv("SS4K_DENSE_MASK")) dense_mask = std::atoi(e);   // A/B switch: which layer pairs of an RDB run fused
  if (const char* e = std::getenv("SS4K_UPS_PRESUM")) ups_presum = e[0] == '1';   // A/B switch: pre-summed weights in the up-sampling convs
  if (const char* e = std::getenv("SS4K_CONV5_MODE")) { conv5_mode = std::atoi(e); use_rs = use_rs || conv5_mode == 1; }   // A/B switch: 0 default, 1 always conv_rs.hip
  if (const char* e = std::getenv("SS4K_NO_RL")) no_rl = e[0] == '1';              // A/B switch: conv5's residual read from memory in the epilogue
  if (const char* e = std::getenv("SS4K_WIDE_RL")) wide_rl = e[0] == '1';          // A/B switch: conv5's residual through the matrix core on the wide kernel
  if (const char* e = std::getenv("SS4K_WIDE")) use_wide = e[0] == '1';            // A/B switch: 64-cout layers on conv3x3_wide_kernel
  if (const char* e = std::getenv("SS4K_W16")) use_w16 = e[0] == '1';              // A/B switch: ... on conv3x3_w16_kernel
  if (const char* e = std::getenv("SS4K_D16")) use_d16 = e[0] == '1';              // A/B switch: fused pairs on conv3x3_d16_kernel
  if (const char* e = std::getenv("SS4K_DENSE_MODE")) dense_mode = std::atoi(e);   // A/B switch: 0 default policy, 1 never, 2 every job
  if (const char* e = std::getenv("SS4K_LANE_GRID")) lane_grid_share = (float)std::atof(e);   // A/B switch: grid of a lane's launch as a share of the chip's slots
  if (const char* e = std::getenv("SS4K_FAIL_AT_CONV")) fail_at_conv = std::atoi(e);   // fault injection: the k-th conv call of every
                                                                                        // other forward throws (tests the unwind of a forked forward)
#endif
  SS4K_REQUIRE(n == model_param_count(desc), "weight blob size does not match the model description");
  ParamCursor pc{w, n};
  if (desc.kind == SS4K_FSRCNN) {
    // The fp16 hi/lo split (fsrcnn.hip split2) needs every operand inside the fp16 range: |x| < 65504 (below 2^-14 the hi part
    // is an fp16 subnormal and the lo part makes up the difference, absolute error < 2^-24 |w|).  Weights are checked here - a
    // checkpoint that does not fit runs the exact-fp32 kernels; activations of an image-range network are orders of magnitude
    // inside (T91: < 120 for inputs in [0,1], tests/test_oracle_golden.py::test_fsrcnn_t91_activation_range), and
    // SS4K_MODEL_FS_EXACT is the caller's switch for a network that is not.
    float wmax = 0.f;
    for (size_t i = 0; i < n; ++i) wmax = std::max(wmax, std::fabs(w[i]));
    if (!(wmax < 6.0e4f)) fs_exact = true;
    // device blob layout: see FsrcnnWeights (glue.h)
    std::vector<float> blob;
    auto push = [&](const std::vector<float>& v) { size_t o = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); while (blob.size() % 4) blob.push_back(0.f); return o; };
    auto conv_tcico = [&](const float* W, int co, int ci, int k2) {  // OIHW -> [tap][ci][co]
      std::vector<float> o((size_t)k2 * ci * co);
      for (int a = 0; a < co; ++a) for (int b = 0; b < ci; ++b) for (int t = 0; t < k2; ++t)
        o[((size_t)t * ci + b) * co + a] = W[((size_t)a * ci + b) * k2 + t];
      return o;
    };
    auto vec = [&](const float* p, int k) { return std::vector<float>(p, p + k); };
    size_t off[32]; int k = 0;
    const float* p;
    p = pc.take(56 * 25); off[k++] = push(conv_tcico(p, 56, 1, 25));
    off[k++] = push(vec(pc.take(56), 56)); off[k++] = push(vec(pc.take(56), 56));
    p = pc.take(12 * 56); off[k++] = push(conv_tcico(p, 12, 56, 1));
    off[k++] = push(vec(pc.take(12), 12)); off[k++] = push(vec(pc.take(12), 12));
    for (int l = 0; l < 4; ++l) {
      p = pc.take(12 * 12 * 9); off[k++] = push(conv_tcico(p, 12, 12, 9));
      off[k++] = push(vec(pc.take(12), 12)); off[k++] = push(vec(pc.take(12), 12));
    }
    p = pc.take(56 * 12); off[k++] = push(conv_tcico(p, 56, 12, 1));
    off[k++] = push(vec(pc.take(56), 56)); off[k++] = push(vec(pc.take(56), 56));
    p = pc.take(56 * 81);  // ConvTranspose weight (C_in=56, C_out=1, 9, 9) -> [tap][cin]
    std::vector<float> wd(81 * 56);
    for (int c = 0; c < 56; ++c) for (int t = 0; t < 81; ++t) wd[(size_t)t * 56 + c] = p[(size_t)c * 81 + t];
    off[k++] = push(wd);
    fsw.b_deconv = *pc.take(1);
    // Matrix-core modes: PReLU(x) = a x + b |x| with a = (1 + s) / 2, b = (1 - s) / 2.  The weights and the bias that PRODUCE a channel are
    // scaled by its a here, so the kernels' accumulators hold y = a x, and PReLU(x) = y + c |y| with c = b / a = (1 - s) / (1 + s) - one fma
    // with a free |.| modifier per value (fsrcnn.hip prelu_mx; rounds 3-6: multiply + max on values carried negated where s > 1).  The
    // slope arrays of the blob hold c.  Needs a > 0, i.e. s > -1, for every channel (real checkpoints: T91 x2 0.0 .. 1.04, x4 up to 9.1);
    // a slope at or below -1 + 1/8 - where c would exceed 15 and the scaled weights lose their bits - sends the model to the exact kernels.
    // (offsets into `blob`: 0 w_feat [25][56], 1 b_feat, 2 a_feat, 3 w_shrink [56][12], 4 b, 5 a, 6 + 3l w_map[l] [9][12][12] (tap, cin, cout),
    //  7 + 3l b, 8 + 3l a, 18 w_expand [12][56], 19 b, 20 a, 21 w_deconv [81][56] (tap, cin))
    struct Act { int w, b, a, channels, count, stride; };   // producing weights: `count` values per channel, `stride` apart, first at w + channel
    const Act acts[7] = {{0, 1, 2, 56, 25, 56}, {3, 4, 5, 12, 56, 12}, {6, 7, 8, 12, 108, 12}, {9, 10, 11, 12, 108, 12}, {12, 13, 14, 12, 108, 12},
                         {15, 16, 17, 12, 108, 12}, {18, 19, 20, 56, 12, 56}};
    if (!fs_exact)
      for (const Act& A : acts)
        for (int c = 0; c < A.channels; ++c) if (!(blob[off[A.a] + c] > -0.875f)) fs_exact = true;
    if (!fs_exact) {   // (both matrix-core modes: fp16 and fp32-grade)
      float* B = blob.data();
      for (const Act& A : acts)
        for (int c = 0; c < A.channels; ++c) {
          const float sl = B[off[A.a] + c], sa = 0.5f * (1.f + sl);
          for (int i = 0; i < A.count; ++i) B[off[A.w] + c + (size_t)i * A.stride] *= sa;
          B[off[A.b] + c] *= sa;
          B[off[A.a] + c] = (1.f - sl) / (1.f + sl);
        }
      fsw.prelu_abs = true;
    }
    upload(fs_blob, blob.data(), blob.size() * 4);
    weight_bytes = blob.size() * 4;
    const float* d = fs_blob.as<float>(); k = 0;
    fsw.w_feat = d + off[k++]; fsw.b_feat = d + off[k++]; fsw.a_feat = d + off[k++];
    fsw.w_shrink = d + off[k++]; fsw.b_shrink = d + off[k++]; fsw.a_shrink = d + off[k++];
    for (int l = 0; l < 4; ++l) { fsw.w_map[l] = d + off[k++]; fsw.b_map[l] = d + off[k++]; fsw.a_map[l] = d + off[k++]; }
    fsw.w_expand = d + off[k++]; fsw.b_expand = d + off[k++]; fsw.a_expand = d + off[k++];
    fsw.w_deconv = d + off[k++];
  } else if (desc.kind == SS4K_RRDBNET) {
    const int nf = desc.num_feat, g = desc.num_grow_ch;
    const int cin0 = 3 * (desc.scale == 2 ? 4 : desc.scale == 1 ? 16 : 1);
    add_conv(pc, nf, cin0, spec_plain(cin0), false);
    for (int b = 0; b < desc.num_block; ++b)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 5; ++c) {
          const int co = c < 4 ? g : nf;
          add_conv(pc, co, nf + c * g, c == 0 ? spec_plain(nf) : spec_concat(nf, c * g), false, /*allow_rs=*/true, /*chainable=*/true);
        }
    for (int i = 0; i < 4; ++i) add_conv(pc, nf, nf, spec_plain(nf), false, /*allow_rs=*/true);
    add_conv(pc, 3, nf, spec_plain(nf), false);
  } else if (desc.kind == SS4K_SRVGG) {
    const int nf = desc.num_feat;
    add_conv(pc, nf, 3, spec_plain(3), true);
    for (int i = 0; i < desc.num_block; ++i) add_conv(pc, nf, nf, spec_plain(nf), true, /*allow_rs=*/true);
    add_conv(pc, 3 * desc.scale * desc.scale, nf, spec_plain(nf), false);
  } else {  // BSVD
    for (int blk = 0; blk < 2; ++blk) {
      const int ci = blk == 0 ? 4 : desc.bsvd_mid_ch, co = blk == 0 ? desc.bsvd_mid_ch : 3;
      auto shapes = bsvd_denblock_shapes(desc.bsvd_chns, ci, co, desc.bsvd_interm_ch);
      for (size_t i = 0; i < shapes.size(); ++i) {
        const int cout = shapes[i].first, cin = shapes[i].second;
        const bool masked = (i == 3 || i == 4 || i == 6 || i == 7 || i == 8 || i == 9 || i == 11 || i == 12);
        const bool ps2 = (i == 10 || i == 13);
        add_conv(pc, cout, cin, masked ? (desc.bsvd_stream ? spec_shifted(cin) : spec_masked(cin)) : spec_plain(cin, ps2 ? 1 : 0), false);
      }
    }
  }
  SS4K_REQUIRE(pc.pos == n, "internal: weight cursor did not consume the blob");
}

// ------------------------------------------------------------------------------------------
void Model::conv(int li, const Tens& in0, const Tens* in1, int N, int H, int W, const ConvOpts& o, hipStream_t st) {
  if (plan_only) return;
#ifdef SS4K_DEV
  if (fail_at_conv > 0 && ++conv_calls == fail_at_conv) throw Error(SS4K_EINVAL, "injected failure (SS4K_FAIL_AT_CONV)");
#endif
  const ConvLayer& L = layers[li];
  ConvArgs a{};
  a.in0 = in0.p; a.in0_plane_bytes = in0.plane_bytes; a.in0_plane0 = in0.plane0; a.nchunks0 = L.nchunks0;
  if (in1) { a.in1 = in1->p; a.in1_plane_bytes = in1->plane_bytes; a.in1_plane0 = in1->plane0; a.nchunks1 = L.nchunks1; }
  SS4K_REQUIRE((in1 != nullptr) == (L.nchunks1 > 0), "internal: conv segment mismatch");
  a.N = N; a.job_n = N; a.H = H; a.W = W; a.ups2 = o.ups2;
  a.wpk = L.w.ptr; a.wrs = L.wrs.ptr; a.w16 = L.w16.ptr; a.rs_wide = L.rs_wide ? 1 : 0; a.bias = L.bias.as<float>(); a.prelu = L.has_prelu ? L.prelu.as<float>() : nullptr; a.prelu_le1 = L.prelu_le1 ? 1 : 0;
  a.act = o.act; a.slope = o.slope; a.alpha = o.alpha; a.gamma = o.gamma;
  if (o.res1) { a.res1 = o.res1->p; a.r1_plane_bytes = o.res1->plane_bytes; a.r1_plane0 = o.res1->plane0; }
  if (o.res2) { a.res2 = o.res2->p; a.r2_plane_bytes = o.res2->plane_bytes; a.r2_plane0 = o.res2->plane0; }
  a.bsvd_resid = o.bsvd_resid;
  a.epi = o.epi; a.out = o.out.p; a.out_plane_bytes = o.out.plane_bytes; a.out_plane0 = o.out.plane0;
  a.cout_real = L.cout_real; a.cout_pad = L.cout_pad;
  a.dbg = dbg; a.dbg_buf = dbg_buf; a.mb_override = mb_override; a.s3 = use_s3 ? 1 : 0; a.wide = use_wide ? 1 : 0; a.ups_presum = ups_presum ? 1 : 0; a.wide_rl = wide_rl ? 1 : 0;
#ifdef SS4K_DEV
  static const bool no_band = std::getenv("SS4K_NO_BAND") && std::getenv("SS4K_NO_BAND")[0] == '1';
  a.no_band = no_band ? 1 : 0;
#endif
  a.reverse = (flip_walk && (launch_parity ^= 1)) ? 1 : 0;
  const double flops = 2.0 * 9.0 * L.cin_real * L.cout_real * (double)H * W * (o.epi == EPI_NHWC_SUB2 ? 0.25 : 1.0);
  if (ctx->prof && !section_open) {   // conv section of this forward: first conv launch ... end of the last one, on the caller's stream
    section = ctx->prof_get_events();
    SS4K_HIP(hipEventRecord(section.a, st));
    section_open = true;
  }
#ifdef SS4K_DEV
  if (chain_rec) {
    a.flops = flops * N;
    chain_record(a, L);
    return;
  }
  // experiment: a 64-cout body layer as two 32-cout groups on the per-launch path (what the chain does to conv5)
  static const bool split64 = std::getenv("SS4K_SPLIT64") && std::getenv("SS4K_SPLIT64")[0] == '1';
  if (split64 && L.wch.ptr && a.cout_pad == 64 && a.epi == EPI_NHWC) { a.wpk = L.wch.ptr; a.wrs = nullptr; a.cout_pad = -64; }
#endif
  // conv5 of an RDB (residual = the conv's own input): on the 64-cout tile with the residual through the matrix core (conv_w16.hip; under
  // SS4K_MODEL_NO_W16 conv_dense.hip's wide kernel), for EVERY job size - two workgroups per CU that co-reside with the fused dense-block
  // launches of the other launch chain.  A frame's bits therefore never depend on the size of the job it arrived in.  (Rounds 2-4 also had a
  // register-stationary kernel for this layer, conv_rs.hip: dev library only since round 5, SS4K_DEV_MODEL_CONV5_RS.)
  if (use_wide && !no_rl && !(a.wrs && conv5_mode == 1) && a.res1 && a.act == ACT_NONE && a.nchunks0 == 4 && a.cout_pad == 64 && !a.ups2 &&
      a.res1 + (size_t)a.r1_plane0 * a.r1_plane_bytes == a.in0 + (size_t)a.in0_plane0 * a.in0_plane_bytes) {
    a.wrs = nullptr; a.wide_rl = 1;
  }
  if (cur_lanes <= 1 || N != cur_n) {
    a.flops = flops * N;
    launch_conv3x3(ctx, a, desc.dtype, st);
    return;
  }
  // frame lanes: lane l takes frames [N*l/2, N*(l+1)/2): lane 0 on the caller's stream, lane 1 on the context's lane stream
  if (!forked) {
    SS4K_HIP(hipEventRecord(ctx->lane_fork(), st));
    SS4K_HIP(hipStreamWaitEvent(ctx->lane_stream(), ctx->lane_fork(), 0));
    forked = true;
  }
  for (int l = 0; l < 2; ++l) {
    a.n0 = N * l / 2; a.N = N * (l + 1) / 2 - a.n0;
    a.grid_share = lane_grid_share;
    a.flops = flops * a.N;
    launch_conv3x3(ctx, a, desc.dtype, l == 0 ? st : ctx->lane_stream());
  }
}

// ---- fused layer pair (conv_pair.hip) ---------------------------------------------------------------------------------
bool Model::conv_pair(int li, const Tens& in0, int N, int H, int W, const ConvOpts& o, hipStream_t st) {
  const ConvLayer& A = layers[li]; const ConvLayer& B = layers[li + 1];
  const bool plain_b = o.epi == EPI_NHWC && o.act == ACT_RELU6 && !o.res1 && !o.res2 && !o.bsvd_resid;
  const bool resid_b = (o.epi == EPI_NHWC || o.epi == EPI_NCHW_F32) && o.act == ACT_NONE && o.res1 && !o.res2 && o.bsvd_resid && o.alpha == 1.f;
  if (!use_pair || desc.dtype != SS4K_F16 || dbg || o.ups2 || !(plain_b || resid_b) || A.nchunks1 || B.nchunks1 || A.has_prelu || B.has_prelu ||
      !conv3x3_pair_eligible(A.nchunks0, A.cout_pad, B.nchunks0, B.cout_pad) || (o.epi == EPI_NCHW_F32 && B.cout_real > 8))
    return false;
  if (plan_only) return true;
#ifdef SS4K_DEV
  if (fail_at_conv > 0 && ++conv_calls == fail_at_conv) throw Error(SS4K_EINVAL, "injected failure (SS4K_FAIL_AT_CONV)");
#endif
  PairArgs a{};
  a.in = in0.p; a.in_plane_bytes = in0.plane_bytes; a.in_plane0 = in0.plane0; a.planes_a = A.nchunks0;
  a.wA = reinterpret_cast<const char*>(A.w.ptr); a.biasA = A.bias.as<float>();
  a.wB = reinterpret_cast<const char*>(B.w.ptr); a.biasB = B.bias.as<float>();
  if (o.res1) { a.res = o.res1->p; a.res_plane_bytes = o.res1->plane_bytes; a.res_plane0 = o.res1->plane0; }
  a.out = o.out.p; a.out_plane_bytes = o.out.plane_bytes; a.out_plane0 = o.out.plane0;
  a.epi = plain_b ? 0 : (o.epi == EPI_NHWC ? 1 : 2);
  a.zero_page = ctx->zero_page();
  a.cout_real = B.cout_real;
  a.H = H; a.W = W;
  const double flops = 2.0 * 9.0 * ((double)A.cin_real * A.cout_real + (double)B.cin_real * B.cout_real) * (double)H * W;
  if (ctx->prof && !section_open) {
    section = ctx->prof_get_events();
    SS4K_HIP(hipEventRecord(section.a, st));
    section_open = true;
  }
  if (cur_lanes <= 1 || N != cur_n) {
    a.n0 = 0; a.N = N; a.flops = flops * N;
    launch_conv3x3_pair(ctx, a, st);
    return true;
  }
  if (!forked) {
    SS4K_HIP(hipEventRecord(ctx->lane_fork(), st));
    SS4K_HIP(hipStreamWaitEvent(ctx->lane_stream(), ctx->lane_fork(), 0));
    forked = true;
  }
  for (int l = 0; l < 2; ++l) {
    a.n0 = N * l / 2; a.N = N * (l + 1) / 2 - a.n0;
    a.grid_share = lane_grid_share;
    a.flops = flops * a.N;
    launch_conv3x3_pair(ctx, a, l == 0 ? st : ctx->lane_stream());
  }
  return true;
}

// ---- fused dense-block layer pair (conv_dense.hip) --------------------------------------------------------------------
bool Model::conv_dense(int li, const Tens& in0, const Tens* in1, int N, int H, int W, float slope, const Tens& out1, const Tens& out2, hipStream_t st) {
  const ConvLayer& A = layers[li]; const ConvLayer& B = layers[li + 1];
  const int pair_bit = (A.nchunks0 + A.nchunks1) <= 4 ? 1 : 2;   // dense_mask: 1 = (conv1, conv2), 2 = (conv3, conv4)
  // default: fused.  Measured on the headline network (tools/env_ab.py SS4K_DENSE_MODE, one box, interleaved): 4-frame jobs + 2.4 %,
  // 2-frame jobs + 3.0 %, one-frame jobs + 11.5 % (nothing else covers their launch boundaries and partly filled rounds of tiles)
  const bool want = dense_mode != 1;
  if (!want || !(dense_mask & pair_bit) || desc.dtype != SS4K_F16 || dbg || chain_rec || A.has_prelu || B.has_prelu || A.nchunks0 != B.nchunks0 ||
      !conv3x3_dense2_eligible(A.nchunks0 + A.nchunks1, A.cout_pad, B.nchunks0 + B.nchunks1, B.cout_pad) ||
      (double)N * H * W * rec() >= 4294967296.0)   // the fused kernel keeps 32-bit byte offsets inside a plane: bigger planes take four launches
    return false;
  if (plan_only) return true;
#ifdef SS4K_DEV
  if (fail_at_conv > 0 && ++conv_calls == fail_at_conv) throw Error(SS4K_EINVAL, "injected failure (SS4K_FAIL_AT_CONV)");
#endif
  SS4K_REQUIRE((in1 != nullptr) == (A.nchunks1 > 0), "internal: conv segment mismatch");
  DenseArgs a{};
  a.in0 = in0.p; a.in0_plane_bytes = in0.plane_bytes; a.in0_plane0 = in0.plane0; a.nchunks0 = A.nchunks0;
  if (in1) { a.in1 = in1->p; a.in1_plane_bytes = in1->plane_bytes; a.in1_plane0 = in1->plane0; a.nchunks1 = A.nchunks1; }
  a.w1 = A.w.as<char>(); a.bias1 = A.bias.as<float>();
  a.w2 = B.w.as<char>(); a.bias2 = B.bias.as<float>();
  a.w16p = A.w16p.ptr ? A.w16p.as<char>() : nullptr;
  a.slope = slope;
  a.out1 = out1.p; a.out1_plane_bytes = out1.plane_bytes; a.out1_plane0 = out1.plane0;
  a.out2 = out2.p; a.out2_plane_bytes = out2.plane_bytes; a.out2_plane0 = out2.plane0;
  a.H = H; a.W = W;
  a.reverse = (flip_walk && (launch_parity ^= 1)) ? 1 : 0;
  const double flops = 2.0 * 9.0 * ((double)A.cin_real * A.cout_real + (double)B.cin_real * B.cout_real) * (double)H * W;
  if (ctx->prof && !section_open) {
    section = ctx->prof_get_events();
    SS4K_HIP(hipEventRecord(section.a, st));
    section_open = true;
  }
  if (cur_lanes <= 1 || N != cur_n) {
    a.n0 = 0; a.N = N; a.flops = flops * N;
#ifdef SS4K_DEV
    if (a.w16p) { launch_conv3x3_d16(ctx, a, st); return true; }
#endif
    launch_conv3x3_dense2(ctx, a, st);
    return true;
  }
  if (!forked) {
    SS4K_HIP(hipEventRecord(ctx->lane_fork(), st));
    SS4K_HIP(hipStreamWaitEvent(ctx->lane_stream(), ctx->lane_fork(), 0));
    forked = true;
  }
  for (int l = 0; l < 2; ++l) {
    a.n0 = N * l / 2; a.N = N * (l + 1) / 2 - a.n0;
    a.grid_share = lane_grid_share;
    a.flops = flops * a.N;
#ifdef SS4K_DEV
    if (a.w16p) { launch_conv3x3_d16(ctx, a, l == 0 ? st : ctx->lane_stream()); continue; }
#endif
    launch_conv3x3_dense2(ctx, a, l == 0 ? st : ctx->lane_stream());
  }
  return true;
}

#ifdef SS4K_DEV
// ---- cross-layer chain (conv_chain.hip; dev library only) -----------------------------------------------------------------
void Model::chain_record(const ConvArgs& a, const ConvLayer& L) {
  SS4K_REQUIRE(desc.dtype == SS4K_F16 && a.epi == EPI_NHWC && !a.bsvd_resid && !a.ups2 && !a.prelu &&
               (a.act == ACT_NONE || a.act == ACT_LRELU) && (a.cout_pad == 32 || a.cout_pad == 64),
               "internal: layer cannot run in a conv chain");
  SS4K_REQUIRE(a.cout_pad == 32 || L.wch.ptr, "internal: 64-cout chain layer without its two-group weights");
  const int nch = a.nchunks0 + a.nchunks1, groups = a.cout_pad / 32;
  SS4K_REQUIRE(nch >= 3, "internal: chain layer with fewer than three K-chunks");
  const char* wbase = a.cout_pad == 32 ? reinterpret_cast<const char*>(a.wpk) : L.wch.as<char>();
  ChainLayerRec rec{(int)chain_items.size(), groups, a.out + (size_t)a.out_plane0 * a.out_plane_bytes,
                    a.out + (size_t)(a.out_plane0 + a.cout_pad / 16) * a.out_plane_bytes, a.flops};
  // which K-chunks wait for what: chain_plan.h (pure host logic, unit-tested on the CPU)
  const int kl = (int)chain_layers.size();
  const unsigned cum_k = (unsigned)chain_items.size();                                  // units per tile of layers < k
  const unsigned cum_km1 = kl >= 1 ? (unsigned)chain_layers[kl - 1].first_item : 0u;    // ... of layers < k - 1
  std::vector<const char*> chunk_planes(nch);
  for (int c = 0; c < nch; ++c)
    chunk_planes[c] = c < a.nchunks0 ? a.in0 + (size_t)(a.in0_plane0 + c) * a.in0_plane_bytes
                                     : a.in1 + (size_t)(a.in1_plane0 + c - a.nchunks0) * a.in1_plane_bytes;
  ChainPrevLayer pvl{};
  if (kl >= 1) pvl = ChainPrevLayer{chain_layers.back().out_lo, chain_layers.back().out_hi, chain_layers.back().nitems};
  const ChainLayerPlan plan = chain_plan_layer(chunk_planes, kl >= 1 ? &pvl : nullptr, a.in0_plane_bytes, cum_k, cum_km1);
  const int newest = plan.newest; const unsigned need_old = plan.need_old, need_new = plan.need_new;
  for (int g = 0; g < groups; ++g) {
    ChainItem it{};
    it.in0 = a.in0; it.in0_plane_bytes = a.in0_plane_bytes; it.in0_plane0 = a.in0_plane0; it.nchunks0 = a.nchunks0;
    it.in1 = a.in1; it.in1_plane_bytes = a.in1_plane_bytes; it.in1_plane0 = a.in1_plane0; it.nchunks1 = a.nchunks1;
    it.wpk = wbase + (size_t)g * nch * 9 * 64 * 16;
    it.bias = a.bias + 32 * g;
    it.act = a.act; it.slope = a.slope; it.alpha = a.alpha; it.gamma = a.gamma;
    it.res1 = a.res1; it.r1_plane_bytes = a.r1_plane_bytes; it.r1_plane0 = a.r1_plane0 + 2 * g;
    it.res2 = a.res2; it.r2_plane_bytes = a.r2_plane_bytes; it.r2_plane0 = a.r2_plane0 + 2 * g;
    it.out = a.out; it.out_plane_bytes = a.out_plane_bytes; it.out_plane0 = a.out_plane0 + 2 * g;
    it.newest = newest; it.need_old = need_old; it.need_new = need_new;
    it.pub_need = g > 0 ? cum_k + (unsigned)g : 0u;   // group g publishes after groups < g of this layer on the same tile
    chain_items.push_back(it);
  }
  chain_layers.push_back(rec);
}

void Model::chain_run(int N, int H, int W, hipStream_t st) {
  chain_rec = false;
  if (chain_items.empty()) return;
  if (!chain_err_host) {
    // the sticky error word: pinned host memory mapped into the device's address space.  A unit that times out ORs into it
    // (system scope); no launch resets it and no copy is involved, so an error can neither be lost nor raced - only
    // check_async_error(), which reports it, clears it
    SS4K_HIP(hipHostMalloc(reinterpret_cast<void**>(&chain_err_host), 64, hipHostMallocMapped | hipHostMallocCoherent));
    __atomic_store_n(chain_err_host, 0u, __ATOMIC_RELAXED);
    SS4K_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&chain_err_dev), chain_err_host, 0));
    SS4K_HIP(hipEventCreateWithFlags(&chain_done, hipEventDisableTiming));
  }
  check_async_error(false);   // an EARLIER forward's chain gave up: reported here at the latest (ss4k_model_check reports it at once)
  const size_t bytes = chain_items.size() * sizeof(ChainItem);
  if (chain_uploaded.size() != chain_items.size() || std::memcmp(chain_uploaded.data(), chain_items.data(), bytes) != 0) {
    // a new job shape (or re-allocated activations): rare, so the upload is allowed to wait - for the previous chain launch, which
    // may still be reading the old table, and for the copy itself (the host vector is reused by the next shape change)
    SS4K_HIP(hipStreamSynchronize(st));
    chain_tab.ensure(bytes);
    chain_uploaded = chain_items;
    SS4K_HIP(hipMemcpy(chain_tab.ptr, chain_uploaded.data(), bytes, hipMemcpyHostToDevice));
  }
  const auto waste = [&](int th) { return (double)((H + th - 1) / th * th) / H; };
  int mb = mb_override ? mb_override : (waste(20) < waste(16) - 1e-9 ? 5 : 4);
#ifdef SS4K_DEV
  if (const char* e = std::getenv("SS4K_CHAIN_MB")) mb = std::atoi(e);
#endif
  ChainArgs ca{};
  ca.items = chain_tab.as<ChainItem>(); ca.nitems = (int)chain_items.size();
  ca.N = N; ca.n0 = 0; ca.H = H; ca.W = W;
  const int ntiles = conv_chain_tiles(N, H, W, mb, &ca.tiles_x, &ca.tiles_y);
  chain_ctl.ensure(conv_chain_ctl_bytes(ntiles));
  ca.ctl = chain_ctl.as<unsigned>();
  ca.zero_page = ctx->zero_page();
  ca.err_sticky = chain_err_dev;
#ifdef SS4K_DEV
  if (const char* e = std::getenv("SS4K_CHAIN_SPIN_LIMIT")) ca.spin_limit = (unsigned)std::atoi(e);   // fault injection: units give up early
  if (const char* e = std::getenv("SS4K_CHAIN_GRID")) ca.grid = std::atoi(e);
  if (const char* e = std::getenv("SS4K_CHAIN_ABL")) ca.abl = std::atoi(e);
#endif
  double flops = 0;
  for (const auto& l : chain_layers) flops += l.flops;
  ProfScope prof(ctx, st, PROF_CONV);
  launch_conv_chain(ctx, ca, mb, st);
  prof.done(flops);
  SS4K_HIP(hipEventRecord(chain_done, st));   // ss4k_model_check(wait) waits for THIS launch before it reads the sticky word
  chain_pending = true;
#ifdef SS4K_DEV
  if (ca.abl == 8) {   // statistics build: how often units started blocked, how long they polled
    SS4K_HIP(hipStreamSynchronize(st));
    std::vector<unsigned> h(4 + ntiles + 1);
    SS4K_HIP(hipMemcpy(h.data(), ca.ctl, h.size() * 4, hipMemcpyDeviceToHost));
    fprintf(stderr, "[chain] grid %d: %d items x %d tiles = %u units: %u blocking starts; workgroup-time spent polling: %.1f us at unit starts, %.1f us inside units (all workgroups together)\n",
            ca.grid, ca.nitems, ntiles, (unsigned)ca.nitems * ntiles, h[2], h[3] / 100.0, h[4 + ntiles] / 100.0);
  }
#endif
}

#endif  // SS4K_DEV

// Asynchronous failures of the chain kernel (a unit gave up waiting: a co-running kernel starved it for seconds, or a defect).
// wait: block until the last chain launch has finished, so that THIS forward's status is known before its output is used.
void Model::check_async_error(bool wait) {
  if (!chain_err_host) return;
  if (wait && chain_pending) { SS4K_HIP(hipEventSynchronize(chain_done)); chain_pending = false; }
  if (__atomic_exchange_n(chain_err_host, 0u, __ATOMIC_ACQ_REL) != 0)
    throw Error(SS4K_EHIP, "conv chain: a work unit timed out waiting for its neighbours; the output of the forward(s) since the "
                           "last successful ss4k_model_check is invalid");
}

// Called at the top of a conv network's forward: one launch chain or two?  Both give bit-identical tensors.
// Two chains win when a launch is short next to its fixed costs (launch boundary, prologue, the partly filled last round
// of tiles): up to 4 frames of 720p per job here; they lose nothing-to-1 % on bigger jobs, and they rely on the two HIP
// streams being served concurrently by the hardware queues.  So, unless forced by SS4K_LANES, the choice is MEASURED per
// (n, h, w) over the shape's first six forwards: calls 0 / 1 warm up two chains / one, calls 2-5 run two, one, one, two (A B B A)
// and are timed with events on the caller's stream (no host synchronisation: the events are polled on later calls); the BETTER
// of each mode's two samples is compared.  Both choices come from misjudgements seen in round 5 (dev library: SS4K_LANE_TUNE_LOG=1):
// a service built a second ago starts on an idle chip whose clock is still ramping, which "two chains first, one chain after"
// (rounds 3-4) charged to the two chains - in A B B A the latest and the earliest timed call are two-chain calls; and a sample can
// be an outlier (31.9 and 54.5 ms for the same job: the host stalled in an allocation while it was enqueueing), which a mean
// turns into a wrong choice for the life of the model (119 instead of 124 frames/s) and a minimum ignores.
// Returns the mode this call runs in: 2 = two launch chains, 1 = one.
int Model::tune_step(std::map<std::tuple<int, int, int>, LaneTune>& tab, int n, int h, int w, hipStream_t st) {
  const auto key = std::make_tuple(n, h, w);
  auto it = tab.find(key);
  if (it == tab.end()) {
    if (tab.size() >= LANE_TUNE_MAX) return 1;   // unmeasured shape past the cap: the plain way
    it = tab.emplace(key, LaneTune{}).first;
  }
  LaneTune* t = &it->second;
  if (t->decided) return t->decided;
  const int k = t->calls++;
  if (k < 6) {
    if (k >= 2) {   // timed
      auto& ev = t->ev[k - 2];
      SS4K_HIP(hipEventCreate(&ev[0])); SS4K_HIP(hipEventCreate(&ev[1]));
      SS4K_HIP(hipEventRecord(ev[0], st));
      tune_timed = ev[1];
    }
    static const int order[6] = {2, 1, 2, 1, 1, 2};
    return order[k];
  }
  bool done = true;
  for (auto& pr : t->ev) done = done && hipEventQuery(pr[1]) == hipSuccess;
  if (done) {
    // (a timed forward that threw left its end event unrecorded: no measurement, stay with the plain way)
    float ms[4] = {0, 0, 0, 0};
    bool ok = true;
    for (int i = 0; i < 4; ++i) ok = ok && hipEventElapsedTime(&ms[i], t->ev[i][0], t->ev[i][1]) == hipSuccess && ms[i] > 0.f;
    if (!ok) (void)hipGetLastError();
    const float ms2 = std::min(ms[0], ms[3]), ms1 = std::min(ms[1], ms[2]);   // calls 2, 5: two chains; 3, 4: one
    // two chains unless one chain was CLEARLY faster: where both were measured carefully (RRDBNet / SRVGG / BSVD, 2-8 frames) two chains win
    // by 4-10 % or lose by at most 1 %, so noisy samples should cost the latter, not the former
    t->decided = !ok ? 1 : (ms1 < 0.97f * ms2 ? 1 : 2);
    t->ms[0] = ms2; t->ms[1] = ms1;
#ifdef SS4K_DEV
    static const bool tune_log = std::getenv("SS4K_LANE_TUNE_LOG") != nullptr;
    if (tune_log) std::fprintf(stderr, "[lanes] kind %d job %d x %d x %d: two chains %.3f %.3f ms, one chain %.3f %.3f ms -> %d\n", desc.kind, n, h, w, ms[0], ms[3], ms[1], ms[2], t->decided);
#endif
    for (auto& pr : t->ev) for (auto& e : pr) { (void)hipEventDestroy(e); e = nullptr; }
    return t->decided;
  }
  return 2;
}

}  // namespace ss4k

// Do two streams run BESIDE each other, and at full launch rate (see ss4k_ctx::lane_check in common.h for why that is in doubt)?  Two
// measurements, timed with events on the first stream between a fork to and a join with the second:
//   PAIR        one 0.2 ms idle kernel on each stream.  Side by side: 0.21-0.22 ms.  Two streams of one hardware queue: 0.4 ms.  The bad
//               pairing below: 0.27-0.34 ms (the second kernel starts late).
//   INTERLEAVE  200 kernels of 1 us on each stream, enqueued alternately.  Normally the host's enqueue rate bounds this (1.0 ms, twice
//               the 0.5 ms of 200 kernels on the caller's stream alone).  On the one bad pairing seen (the process's 5th hardware
//               queue against the NULL stream's, GPU_MAX_HW_QUEUES=8) it takes 5.6 ms - 14 us per launch while both queues are busy,
//               which is what cost a 351-launch forward its 6 % (profiles/earlier/r05/r05_lane_queue.txt).
namespace ss4k {
bool stream_pair_ok(hipStream_t caller, hipStream_t ls) {
  static const bool log = std::getenv("SS4K_LANE_CHECK_LOG") != nullptr;
  constexpr unsigned TICKS = 20000;   // 0.2 ms
  constexpr int BURST = 200;
  struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr, fork = nullptr, done = nullptr;
    Events() {
      for (hipEvent_t* e : {&e0, &e1}) if (hipEventCreate(e) != hipSuccess) { drop(); throw Error(SS4K_EHIP, "hipEventCreate failed"); }
      for (hipEvent_t* e : {&fork, &done}) if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) { drop(); throw Error(SS4K_EHIP, "hipEventCreateWithFlags failed"); }
    }
    void drop() { for (hipEvent_t* e : {&e0, &e1, &fork, &done}) if (*e) { (void)hipEventDestroy(*e); *e = nullptr; } }
    ~Events() { drop(); }
  } ev;
  // the elapsed time on the caller's stream of `body`, run between a fork to and a join with the other stream
  auto timed = [&](auto&& body) {
    float ms = 0.f;
    SS4K_HIP(hipEventRecord(ev.fork, caller));
    SS4K_HIP(hipStreamWaitEvent(ls, ev.fork, 0));
    SS4K_HIP(hipEventRecord(ev.e0, caller));
    body();
    SS4K_HIP(hipEventRecord(ev.done, ls));
    SS4K_HIP(hipStreamWaitEvent(caller, ev.done, 0));
    SS4K_HIP(hipEventRecord(ev.e1, caller));
    SS4K_HIP(hipEventSynchronize(ev.e1));
    SS4K_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    return ms;
  };
  op_lane_spin(100, caller);                      // (the first launch of the kernel carries its load time)
  const float one = timed([&] { op_lane_spin(TICKS, caller); });
  float p3[3];
  for (float& v : p3) v = timed([&] { op_lane_spin(TICKS, ls); op_lane_spin(TICKS, caller); });
  std::sort(p3, p3 + 3);
  const float pair = p3[1];   // the median: bad pairings showed 0.27-0.34 ms with a stray 0.23, good ones 0.21-0.22 with a stray 0.24
  const float burst_one = timed([&] { for (int i = 0; i < BURST; ++i) op_lane_spin(100, caller); });
  const float burst_both = timed([&] { for (int i = 0; i < BURST; ++i) { op_lane_spin(100, caller); op_lane_spin(100, ls); } });
  const bool ok = pair - one < 0.045f && burst_both < 5.f * burst_one;   // (either sign alone has missed a bad pairing once)
  if (log) std::fprintf(stderr, "[streams] %p and %p: one 0.2 ms kernel %.3f ms, one on each %.3f ms; %d x 1 us kernels: %.3f ms on the first stream, %.3f ms interleaved -> %s\n",
                        (void*)caller, (void*)ls, one, pair, BURST, burst_one, burst_both, ok ? "side by side" : "NOT side by side");
  return ok;
}
}  // namespace ss4k

void ss4k_ctx::lane_check(hipStream_t caller) {
  if (lane_checked.count(caller)) return;
  // SS4K_NO_LANE_CHECK (include/ss4k.h, INTEGRATION.md "Runtime settings") switches the test off; so does a profiler that has loaded itself into
  // the process (rocprofv3 sets ROCP_TOOL_LIBRARIES / preloads librocprofiler-sdk): counter collection serialises kernels, every pair
  // "fails", and six parked streams later the last one is used untested anyway
  static const bool off = [] {
    if (std::getenv("SS4K_NO_LANE_CHECK")) return true;
    if (std::getenv("ROCP_TOOL_LIBRARIES")) return true;
    const char* pre = std::getenv("LD_PRELOAD");
    return pre && (std::strstr(pre, "rocprofiler") || std::strstr(pre, "roctracer"));
  }();
  if (off) return;
  // (a stream that is being captured into a graph cannot be synchronised: the test waits for the first forward outside a capture)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(caller, &cap) != hipSuccess) { (void)hipGetLastError(); return; }
  if (cap != hipStreamCaptureStatusNone) return;
  lane_checked.insert(caller);
  for (int attempt = 0; attempt < 6; ++attempt) {
    // ... and beside the NULL stream: the usual place for a host to WAIT for the caller's stream (torch's current stream in a worker that runs
    // its jobs on side streams), and a queue that only waits slows its slow partner just as a busy one does
    // The NULL-stream half launches on the legacy stream: skipped when the caller IS that stream, and when some OTHER stream of the process
    // is under a global-mode graph capture (a legacy-stream launch would invalidate that capture: the runtime reports it as an error of
    // hipStreamIsCapturing on the NULL stream).
    bool null_probe = caller != nullptr;
    if (null_probe) {
      hipStreamCaptureStatus ncap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(nullptr, &ncap) != hipSuccess) { (void)hipGetLastError(); null_probe = false; }
      else if (ncap != hipStreamCaptureStatusNone) null_probe = false;
    }
    if (ss4k::stream_pair_ok(caller, lane_stream()) && (!null_probe || ss4k::stream_pair_ok(nullptr, lane_stream()))) break;
    lane_parked.push_back(lane_stream_); lane_stream_ = nullptr; ++lane_replaced;   // (after 6 tries the last new stream is used untested)
  }
}

namespace ss4k {

void Model::lanes_begin(int n, int h, int w, hipStream_t st) {
  cur_lanes = 1; cur_n = n; forked = false; tune_timed = nullptr;
  if (plan_only || lanes_mode == 1 || n < 2 || desc.dtype != SS4K_F16 || dbg) return;   // (an odd job splits 1 : 2, 2 : 3, ...)
  ctx->lane_check(st);   // (first job from this stream only)
  if (lanes_mode == 2) { cur_lanes = 2; return; }
  cur_lanes = tune_step(lane_tune, n, h, w, st);
}

// the caller's stream continues only after both chains: called before any non-conv work on the tensors and at the end
void Model::lanes_join(hipStream_t st, bool end_of_forward) {
  if (forked) {
    SS4K_HIP(hipEventRecord(ctx->lane_done(), ctx->lane_stream()));
    SS4K_HIP(hipStreamWaitEvent(st, ctx->lane_done(), 0));
    forked = false;
  }
  if (section_open) {
    SS4K_HIP(hipEventRecord(section.b, st));
    ctx->prof_sections.push_back(section);
    section_open = false;
  }
  if (end_of_forward && tune_timed) { SS4K_HIP(hipEventRecord(tune_timed, st)); tune_timed = nullptr; }
}

size_t Model::workspace_bytes(int n, int h, int w) {
  plan_only = true; plan_bytes.clear();
  try { forward(nullptr, nullptr, n, h, w, nullptr); } catch (...) { plan_only = false; throw; }
  plan_only = false;
  size_t total = 0;
  for (size_t b : plan_bytes) total += (b + 255) & ~size_t(255);
  return total;
}

// activation buffer idx holding `channels` channels (rounded up to whole 32-cout groups of planes,
// which is what a producing conv writes) for `pixels` pixels
Tens Model::act(int idx, size_t pixels, int channels) {
  if ((int)acts.size() <= idx) acts.resize(idx + 1);
  // what a producing conv writes: one 32-cout block up to 32 channels, whole 64-cout groups beyond (pack_conv3x3's cout_pad - a
  // 96-channel tensor is written as 128: sized to 96 its last two planes would land past the buffer)
  const int ch32 = channels <= 32 ? 32 : (channels + 63) / 64 * 64;
  const int planes = planes_for(ch32);
  SS4K_REQUIRE(pixels < 2147483648ull, "an activation plane holds at most 2^31 pixels");
  const size_t need = (size_t)planes * pixels * rec();
  if (plan_only) {
    if ((int)plan_bytes.size() <= idx) plan_bytes.resize(idx + 1, 0);
    plan_bytes[idx] = std::max(plan_bytes[idx], need);
    return Tens{nullptr, pixels * (size_t)rec(), 0};
  }
  acts[idx].ensure(need);
  return Tens{acts[idx].as<char>(), pixels * (size_t)rec(), 0};
}

void Model::out_shape(int n, int h, int w, int* oc, int* oh, int* ow) const {
  (void)n;
  switch (desc.kind) {
    case SS4K_FSRCNN: *oc = 1; *oh = h * desc.scale; *ow = w * desc.scale; break;
    case SS4K_RRDBNET: *oc = 3; *oh = h * desc.scale; *ow = w * desc.scale; break;
    case SS4K_SRVGG: *oc = 3; *oh = h * desc.scale; *ow = w * desc.scale; break;
    default: *oc = 3; *oh = h; *ow = w; break;
  }
}
int Model::in_channels() const {
  return desc.kind == SS4K_FSRCNN ? 1 : desc.kind == SS4K_BSVD ? 4 : 3;
}

void Model::pack_in(const float* in, const Tens& dst, int nplanes, int n, int c, int h, int w, int r, hipStream_t st) {
  if (plan_only) return;
  if (desc.dtype == SS4K_F16) op_pack_input<__half>(in, reinterpret_cast<__half*>(dst.p), n, c, h, w, r, nplanes, st);
  else op_pack_input<float>(in, reinterpret_cast<float*>(dst.p), n, c, h, w, r, nplanes, st);
}

void Model::abort_forward(hipStream_t st) noexcept {
  if (forked) {
    // lane-1 kernels may still be running on the context's lane stream: the caller's stream waits for them
    hipEvent_t ev = nullptr;
    try { ev = ctx->lane_done(); } catch (...) { ev = nullptr; }
    if (!ev || hipEventRecord(ev, ctx->lane_stream_) != hipSuccess || hipStreamWaitEvent(st, ev, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (ctx->lane_stream_) (void)hipStreamSynchronize(ctx->lane_stream_);   // last resort: block the host
    }
    forked = false;
  }
  if (section_open) { ctx->prof_pool.push_back(section); section_open = false; }
  tune_timed = nullptr; cur_lanes = 1;
  chain_rec = false;
  out_stats_acc = nullptr; out_stats_done = false; out_half = false; in_u8 = false;
}

void Model::forward(const float* in, float* out, int n, int h, int w, hipStream_t st) {
  try { forward_impl(in, out, n, h, w, st); }
  catch (...) { if (!plan_only) abort_forward(st); throw; }
}

void Model::forward_impl(const float* in, float* out, int n, int h, int w, hipStream_t st) {
  SS4K_REQUIRE(n > 0 && h > 0 && w > 0, "forward: empty input");
  out_stats_done = false;
  const bool f16 = desc.dtype == SS4K_F16;
  if (desc.kind != SS4K_FSRCNN) {
    // the conv kernel indexes the pixels of a plane with 32 bits: split batches whose largest internal
    // tensor would exceed 2^31 pixels (none of the BASELINE shapes: 4 frames at 4320x7680 are 133 M)
    int oc, oh, ow; out_shape(1, h, w, &oc, &oh, &ow);
    // largest grid a "planes" tensor of this network lives on: RRDBNet's tail runs at the output
    // resolution; SRVGG (PixelShuffle in the glue tail) and BSVD never leave the input resolution
    const bool tail_at_out = desc.kind == SS4K_RRDBNET;
    const double plane1 = (double)(tail_at_out ? std::max(oh, h) : h) * (tail_at_out ? std::max(ow, w) : w);
    SS4K_REQUIRE(plane1 < 2147483648.0, "forward: a single frame exceeds 2^31 pixels");
    int max_n = std::max(1, (int)(2147483647.0 / plane1));
    if (sub_batch > 0) max_n = std::min(max_n, sub_batch);
    if (n > max_n) {
      out_stats_acc = nullptr;   // per-plane accumulation is not offered across a split batch: the caller makes its own pass
      SS4K_REQUIRE(!out_half, "forward: an fp16 output tensor is not offered across a split batch");
      for (int i = 0; i < n; i += max_n) {
        const int nn = std::min(max_n, n - i);
        forward(in + (size_t)i * in_channels() * h * w, out + (size_t)i * oc * oh * ow, nn, h, w, st);
      }
      return;
    }
  }
  if (desc.kind == SS4K_FSRCNN) {
    const size_t px = (size_t)n * h * w;
    if (acts.size() < 2) acts.resize(2);
    if (plan_only) { plan_bytes.assign(2, px * 12 * 4); return; }
    acts[0].ensure(px * 12 * 4); acts[1].ensure(px * 12 * 4);
    const bool half_out = out_half; out_half = false;
    const bool u8_in = in_u8; in_u8 = false;
    fsrcnn_forward(ctx, fsw, desc.scale, in, out, n, h, w, acts[0].as<float>(), acts[1].as<float>(),
                   fs_exact ? FS_MODE_EXACT : f16 ? FS_MODE_HALF : FS_MODE_SPLIT, st, half_out, u8_in);
    return;
  }
  lanes_begin(n, h, w, st);
#ifdef SS4K_DEV
  conv_calls = (forward_calls++ & 1) ? -(1 << 30) : 0;   // fault injection (SS4K_FAIL_AT_CONV): every other forward fails
#endif
  // plane index of channel c inside a tensor
  auto plane_of = [&](const Tens& t, int channel) { return Tens{t.p, t.plane_bytes, t.plane0 + channel / cw()}; };
  auto nchw_out = [&]() { return Tens{reinterpret_cast<char*>(out), 0, 0}; };
  if (desc.kind == SS4K_RRDBNET) {
    const int nf = desc.num_feat, g = desc.num_grow_ch;
    const int r = desc.scale == 2 ? 2 : desc.scale == 1 ? 4 : 1;
    SS4K_REQUIRE(h % r == 0 && w % r == 0, "RRDBNet: input size must be divisible by the pixel-unshuffle factor");
    const int H = h / r, W = w / r;
    const size_t px = (size_t)n * H * W;
    const int cin0 = 3 * r * r;
    Tens P = act(0, px, cin0);
    Tens F = act(1, px, nf), X[3] = {act(2, px, nf), act(3, px, nf), act(4, px, nf)};
    Tens G = act(5, px, 4 * g);
    pack_in(in, P, planes_for(cin0), n, 3, h, w, r, st);
    int li = 0;
    { ConvOpts o; o.out = F; conv(li++, P, nullptr, n, H, W, o, st); }
    Tens cur = F;
    // the body of an fp16 job as ONE persistent launch with per-tile hand-offs between the layers (conv_chain.hip): opt-in
    // (SS4K_MODEL_CHAIN), never the default.  Measured in round 3 on one 720p frame (460 tiles per layer, 512 workgroup slots): the chain runs a single
    // caller's 1-frame jobs 1-6 % faster than 345 launches (box by box), but two callers alternating on two streams are better
    // off with launches (105-112 against 98 frames/s: their chains of launches fill each other's gaps, two chain kernels only
    // compete for the slots) - and a model cannot know how many callers the GPU has; since round 4 the fused dense-block launches
    // (conv_dense.hip) give the single caller more than the chain does.  DESIGN.md 4.5, profiles/NOTES_r01_r03.md 4.1d.
#ifdef SS4K_DEV
    const bool use_chain = !plan_only && f16 && !dbg && nf == 64 && g == 32 && chain_mode == 2;
    if (use_chain) {
      chain_rec = true; chain_items.clear(); chain_layers.clear();
    }
#endif
    for (int b = 0; b < desc.num_block; ++b) {
      const Tens a = cur;
      const Tens t1 = X[0], t2 = X[1];
      const Tens dst = (a.p == F.p) ? X[2] : a;
      const Tens rin[3] = {a, t1, t2};
      const Tens rout[3] = {t1, t2, dst};
      for (int rr = 0; rr < 3; ++rr) {
        for (int c = 0; c < 4; ++c) {
          // (conv1, conv2) and (conv3, conv4) read the same planes: one fused launch each where the shape fits (conv_dense.hip)
          if (!(c & 1) && conv_dense(li, rin[rr], c == 0 ? nullptr : &G, n, H, W, 0.2f, plane_of(G, c * g), plane_of(G, (c + 1) * g), st)) {
            li += 2; ++c;
            continue;
          }
          ConvOpts o; o.act = ACT_LRELU; o.slope = 0.2f; o.out = plane_of(G, c * g);
          conv(li++, rin[rr], c == 0 ? nullptr : &G, n, H, W, o, st);
        }
        ConvOpts o; o.alpha = 0.2f; o.res1 = &rin[rr]; o.out = rout[rr];
        if (rr == 2) { o.gamma = 0.2f; o.res2 = &a; }
        conv(li++, rin[rr], &G, n, H, W, o, st);
      }
      cur = dst;
    }
#ifdef SS4K_DEV
    if (use_chain) {
      if (forked) {   // (a forced chain on an even batch: conv_first ran as two launch chains; the conv section stays open)
        SS4K_HIP(hipEventRecord(ctx->lane_done(), ctx->lane_stream()));
        SS4K_HIP(hipStreamWaitEvent(st, ctx->lane_done(), 0));
        forked = false;
      }
      cur_lanes = 1;
      chain_run(n, H, W, st);
    }
#endif
    { ConvOpts o; o.res1 = &F; o.out = X[0]; conv(li++, cur, nullptr, n, H, W, o, st); }  // feat + conv_body(body)
    Tens U1 = act(6, px * 4, nf), U2 = act(7, px * 16, nf), U3 = act(8, px * 16, nf);
    { ConvOpts o; o.ups2 = 1; o.act = ACT_LRELU; o.slope = 0.2f; o.out = U1; conv(li++, X[0], nullptr, n, 2 * H, 2 * W, o, st); }
    { ConvOpts o; o.ups2 = 1; o.act = ACT_LRELU; o.slope = 0.2f; o.out = U2; conv(li++, U1, nullptr, n, 4 * H, 4 * W, o, st); }
    { ConvOpts o; o.act = ACT_LRELU; o.slope = 0.2f; o.out = U3; conv(li++, U2, nullptr, n, 4 * H, 4 * W, o, st); }
    { ConvOpts o; o.epi = EPI_NCHW_F32; o.out = nchw_out(); conv(li++, U3, nullptr, n, 4 * H, 4 * W, o, st); }
    lanes_join(st, true);
    return;
  }
  if (desc.kind == SS4K_SRVGG) {
    const int nf = desc.num_feat;
    const size_t px = (size_t)n * h * w;
    Tens P = act(0, px, 3), A = act(1, px, nf), B = act(2, px, nf);
    pack_in(in, P, planes_for(3), n, 3, h, w, 1, st);
    int li = 0;
    { ConvOpts o; o.act = ACT_PRELU; o.out = A; conv(li++, P, nullptr, n, h, w, o, st); }
    Tens cur = A, nxt = B;
    for (int i = 0; i < desc.num_block; ++i) {
      ConvOpts o; o.act = ACT_PRELU; o.out = nxt; conv(li++, cur, nullptr, n, h, w, o, st);
      std::swap(cur, nxt);
    }
    Tens Z = act(3, px, layers[li].cout_pad);
    { ConvOpts o; o.out = Z; conv(li++, cur, nullptr, n, h, w, o, st); }
    lanes_join(st, true);
    if (plan_only) return;
    // the service may ask for the output's plane statistics to be accumulated while it is written
    double* sacc = out_stats_acc; out_stats_acc = nullptr; out_stats_done = sacc != nullptr;
    const bool half_out = out_half; out_half = false;
    SS4K_REQUIRE(!half_out || f16, "internal: fp16 output tensor requested from an fp32 network");
    if (half_out) op_ps_nchw_addbase<__half, __half>(reinterpret_cast<const __half*>(Z.p), reinterpret_cast<__half*>(out), in, n, h, w, desc.scale, 3, sacc, st);
    else if (f16) op_ps_nchw_addbase<__half>(reinterpret_cast<const __half*>(Z.p), out, in, n, h, w, desc.scale, 3, sacc, st);
    else op_ps_nchw_addbase<float>(reinterpret_cast<const float*>(Z.p), out, in, n, h, w, desc.scale, 3, sacc, st);
    return;
  }
  // ---- BSVD, one frame per call -----------------------------------------------------------
  SS4K_REQUIRE(h % 4 == 0 && w % 4 == 0, "BSVD: frame size must be divisible by 4");
  const int c0 = desc.bsvd_chns[0], c1 = desc.bsvd_chns[1], c2 = desc.bsvd_chns[2], mid = desc.bsvd_mid_ch;
  const size_t px = (size_t)n * h * w, px2 = px / 4, px4 = px / 16;
  const int h2 = h / 2, w2 = w / 2, h4 = h / 4, w4 = w / 4;
  Tens IN0 = act(0, px, 4), MID = act(1, px, mid);
  pack_in(in, IN0, planes_for(4), n, 4, h, w, 1, st);
  int li = 0;
  for (int blk = 0; blk < 2; ++blk) {
    const Tens IN = blk == 0 ? IN0 : MID;
    Tens X0 = act(3, px, c0);   // (I0 and O0, the tensors inside the inc / outc pairs, exist only when the pairs run as two launches)
    Tens D0 = act(4, px2, c1), Ma = act(5, px2, c1), X1 = act(6, px2, c1);
    Tens D1 = act(7, px4, c2), Mb = act(8, px4, c2), X2 = act(9, px4, c2), Mc = act(10, px4, c2);
    Tens S1 = act(11, px2, c1), S0 = act(12, px, c0);
    auto relu6 = [&](Tens outT) { ConvOpts o; o.act = ACT_RELU6; o.out = outT; return o; };
    // masked conv input: skip the planes that only hold dead channels (see spec_masked)
    auto masked = [&](const Tens& t, int c) { return Tens{t.p, t.plane_bytes, t.plane0 + (c / 4) / cw()}; };
    // one BiBufferConv + ReLU6: independent frames read only the live planes; in a stream the
    // leading planes are first rebuilt from the neighbouring frames (op_temporal_shift)
    auto bibuf = [&](const Tens& t, int c, int N, int H, int W, const Tens& outT) {
      if (!desc.bsvd_stream) { conv(li++, masked(t, c), nullptr, N, H, W, relu6(outT), st); return; }
      const int lead = shifted_planes(c);
      lanes_join(st, false);   // the time shift reads neighbouring frames
      Tens S = act(14, (size_t)N * H * W, lead * cw());
      if (!plan_only) op_temporal_shift(t.p + (size_t)t.plane0 * t.plane_bytes, S.p, lead, N, (size_t)H * W, rec() / 16, cw(), c / 8, st);
      const Tens rest{t.p, t.plane_bytes, t.plane0 + lead};
      conv(li++, S, planes_for(c) > lead ? &rest : nullptr, N, H, W, relu6(outT), st);
    };
    // (workspace planning counts the tensors inside the inc / outc pairs whichever route runs: a pair that falls back to two
    // launches at run time - a measurement selector set after the query - must never find the workspace under-reported)
    if (plan_only) { (void)act(2, px, desc.bsvd_interm_ch); (void)act(13, px, c0); }
    if (conv_pair(li, IN, n, h, w, relu6(X0), st)) li += 2;                            // inc.convblock.0 + .3 fused (conv_pair.hip)
    else {
      Tens I0 = act(2, px, desc.bsvd_interm_ch);
      conv(li++, IN, nullptr, n, h, w, relu6(I0), st);                                 // inc.convblock.0
      conv(li++, I0, nullptr, n, h, w, relu6(X0), st);                                 // inc.convblock.3
    }
    { ConvOpts o = relu6(D0); o.epi = EPI_NHWC_SUB2; conv(li++, X0, nullptr, n, h, w, o, st); }   // downc0 stride 2
    bibuf(D0, c1, n, h2, w2, Ma);
    bibuf(Ma, c1, n, h2, w2, X1);
    { ConvOpts o = relu6(D1); o.epi = EPI_NHWC_SUB2; conv(li++, X1, nullptr, n, h2, w2, o, st); } // downc1 stride 2
    bibuf(D1, c2, n, h4, w4, Mb);
    bibuf(Mb, c2, n, h4, w4, X2);
    bibuf(X2, c2, n, h4, w4, Mc);                     // upc2.memconv
    bibuf(Mc, c2, n, h4, w4, Mb);
    { ConvOpts o; o.epi = EPI_NHWC_PS2; o.res1 = &X1; o.out = S1; conv(li++, Mb, nullptr, n, h4, w4, o, st); }  // PixelShuffle + skip3
    bibuf(S1, c1, n, h2, w2, Ma);                     // upc1.memconv
    bibuf(Ma, c1, n, h2, w2, D0);
    { ConvOpts o; o.epi = EPI_NHWC_PS2; o.res1 = &X0; o.out = S0; conv(li++, D0, nullptr, n, h2, w2, o, st); }  // PixelShuffle + skip2
    { ConvOpts o; o.res1 = &IN; o.bsvd_resid = 1;
      if (blk == 0) { o.out = MID; } else { o.epi = EPI_NCHW_F32; o.out = nchw_out(); }
      if (conv_pair(li, S0, n, h, w, o, st)) li += 2;                                  // outc.convblock.0 + .3 + residual fused
      else {
        Tens O0 = act(13, px, c0);
        conv(li++, S0, nullptr, n, h, w, relu6(O0), st);                               // outc.convblock.0
        conv(li++, O0, nullptr, n, h, w, o, st);                                       // outc.convblock.3 + residual
      } }
  }
  lanes_join(st, true);
}


#ifdef SS4K_DEV
// ---- measurement hook: one conv layer in isolation (ss4k_bench_conv) --------------------------
double bench_conv_layer(ss4k_ctx* ctx, int dtype, int cin0, int cin1, int cout, int n, int h, int w, int flags,
                        int iters, hipStream_t st) {
  ss4k_model_desc d{}; d.kind = SS4K_RRDBNET; d.dtype = dtype; d.scale = 2; d.num_feat = 64; d.num_block = 1; d.num_grow_ch = 32;
  Model m; m.ctx = ctx; m.desc = d;
  m.use_rs = (flags & 4096) != 0;   // 4096: the register-stationary kernel (conv_rs.hip) where the shape is built
  m.conv5_mode = m.use_rs ? 1 : 0;
  m.rs_mask = 63; m.rs_wide = (flags & 8192) != 0;
  const int cin = cin0 + cin1;
  std::vector<float> blob((size_t)cout * cin * 9 + cout);
  uint32_t s = 12345;
  for (auto& v : blob) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f * 0.02f - 0.01f; }
  ParamCursor pc{blob.data(), blob.size()};
  const int li = m.add_conv(pc, cout, cin, cin1 ? m.spec_concat(cin0, cin1) : m.spec_plain(cin0), false, /*allow_rs=*/true);
  const size_t px = (size_t)n * h * w;
  Tens X = m.act(0, px, cin0), G = m.act(1, px, std::max(cin1, 32)), O = m.act(2, px, cout);
  {  // random operands: constant data lets the chip hold a higher clock than real frames do
    auto fill = [&](Tens& t, int ch) {
      const size_t bytes = (size_t)m.planes_for(ch) * px * m.rec();
      std::vector<uint32_t> hbuf(bytes / 4);
      for (auto& v : hbuf) {
        s = s * 1664525u + 1013904223u;
        v = dtype == SS4K_F16 ? ((s & 0x83FF83FFu) | 0x38003800u)    // two halves in +-[0.5,1)
                              : ((s & 0x807FFFFFu) | 0x3F000000u);  // one float in +-[0.5,1)
      }
      SS4K_HIP(hipMemcpy(t.p, hbuf.data(), bytes, hipMemcpyHostToDevice));
    };
    fill(X, cin0); fill(G, std::max(cin1, 32));
  }
  ConvOpts o; o.act = ACT_LRELU; o.slope = 0.2f; o.out = O;
  if ((flags & 2048) && cout <= cin0) { o.act = ACT_NONE; o.alpha = 0.2f; o.res1 = &X; }  // conv5 of an RDB: x5 * 0.2 + x
  m.dbg = flags & ~(2048 | 4096 | 8192);
  DevBuf dbgb; dbgb.ensure(1024 * 16 * 8); SS4K_HIP(hipMemsetAsync(dbgb.ptr, 0, 1024 * 16 * 8, st)); m.dbg_buf = dbgb.as<unsigned long long>();
  for (int i = 0; i < 3; ++i) m.conv(li, X, cin1 ? &G : nullptr, n, h, w, o, st);
  hipEvent_t e0, e1; SS4K_HIP(hipEventCreate(&e0)); SS4K_HIP(hipEventCreate(&e1));
  SS4K_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) m.conv(li, X, cin1 ? &G : nullptr, n, h, w, o, st);
  SS4K_HIP(hipEventRecord(e1, st));
  SS4K_HIP(hipEventSynchronize(e1));
  float ms = 0; SS4K_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (flags & DBG_STAMP) {  // print the phase breakdown of the last launch (wave 0 of every workgroup)
    std::vector<unsigned long long> h((size_t)1024 * 16);
    SS4K_HIP(hipMemcpy(h.data(), m.dbg_buf, h.size() * 8, hipMemcpyDeviceToHost));
    double v[16] = {0}; int nwg = 0;
    for (int i = 0; i < 1024; ++i) if (h[i * 16 + 5]) { for (int k = 0; k < 16; ++k) v[k] += (double)h[i * 16 + k]; ++nwg; }
    if (nwg) {
      unsigned long long b0 = ~0ull, b1 = 0, e0 = ~0ull, e1 = 0; double tmax = 0, tmin = 1e30; int tl_max = 0;
      for (int i = 0; i < 1024; ++i) if (h[i * 16 + 5]) {
        b0 = std::min(b0, h[i * 16 + 14]); b1 = std::max(b1, h[i * 16 + 14]);
        e0 = std::min(e0, h[i * 16 + 15]); e1 = std::max(e1, h[i * 16 + 15]);
        tmax = std::max(tmax, (double)h[i * 16]); tmin = std::min(tmin, (double)h[i * 16]); tl_max = std::max(tl_max, (int)h[i * 16 + 5]);
      }
      fprintf(stderr, "[stamp] kernel span %.1f us: workgroup starts spread over %.1f us, ends over %.1f us; per-WG cycles min %.0f max %.0f, max tiles %d\n",
              (e1 - b0) / 100.0, (b1 - b0) / 100.0, (e1 - e0) / 100.0, tmin, tmax, tl_max);
      for (double& x : v) x /= nwg;
      fprintf(stderr, "[stamp] %d WGs, avg tiles %.2f, clock %.0f MHz; wave0 cycles: total %.0f  issue %.0f  mma %.0f  epilogue %.0f  dma-wait %.0f  barrier %.0f  prologue %.0f | last wave: mma %.0f  epilogue %.0f  dma-wait %.0f  barrier %.0f\n",
              nwg, v[5], v[9] > 0 ? v[0] / v[9] * 100.0 : 0.0, v[0], v[1], v[2], v[3], v[8], v[4], v[6], v[10], v[13], v[12], v[11]);
    }
  }
  dbgb.release();
  return 1000.0 * ms / iters;
}

#endif  // SS4K_DEV

}  // namespace ss4k
