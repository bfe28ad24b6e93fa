// C ABI (include/ss4k.h): context, models, the frame-in/frame-out upscaler and the granular ops.
#include "models.h"
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <memory>

namespace ss4k {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}

template <typename F>
static int guard(F&& f) {
  try { f(); g_err[0] = 0; return SS4K_OK; }
  catch (const Error& e) { set_error("%s", e.what()); return e.code; }
  catch (const std::bad_alloc&) { set_error("out of host memory"); return SS4K_ENOMEM; }
  catch (const std::exception& e) { set_error("%s", e.what()); return SS4K_EINVAL; }
}

// blur_ker (fsrcnn_upscaler.py:20-52) as its 1-D factor: the reference's normalised 2-D kernel
// (1 / (2 pi var)) exp(-(dx^2 + dy^2) / (2 var)) / sum is g[y] * g[x] with g = e / sum(e)
static std::vector<float> gaussian_taps_1d(int k, float sigma) {
  std::vector<float> g(k);
  const float mean = (k - 1) / 2.0f, var = sigma * sigma;
  double sum = 0.0;
  for (int i = 0; i < k; ++i) { const float d = i - mean; g[i] = expf(-(d * d) / (2 * var)); sum += g[i]; }
  for (auto& v : g) v = (float)(v / sum);
  return g;
}
static std::vector<float> sharpen_taps(double strength) {  // sharpen_ker, fsrcnn_upscaler.py:54-84
  std::vector<float> t(9);
  const float s = (float)strength, one_m = (float)(1.0 - strength);
  float sum = 0.f;
  for (int i = 0; i < 9; ++i) {
    const float sharp = i == 4 ? 9.f : -1.f, ident = i == 4 ? 1.f : 0.f;
    t[i] = sharp * s + one_m * ident; sum += t[i];
  }
  for (auto& v : t) v /= sum;
  return t;
}

struct Upscaler {
  ss4k_ctx* ctx; ss4k_upscale_cfg cfg; Model* sr; Model* dn;
  DevBuf k_gauss17, k_sharp, k_sharp_hr;
  DevBuf img, lr, lr4, den, hr, hr2, lb, hb, lbb, hbb, st_hr, st_lr, st_acc, st_acc2;
  bool acc2_clean = false;   // st_acc2 holds zeros (its last user re-zeroed what it had summed: k_stats_final2)
  bool first_frame = true;
  bool taps_on = false;
  // host time spent enqueueing the last job's denoise / SR model stages: what the reference's
  // 'fsrcnn.denoise' / 'fsrcnn.model' profiler spans measure on an asynchronous device queue
  // (util/profiler.py:12-24 - no device sync)
  double enq_denoise_ms = 0, enq_model_ms = 0;
  static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  DevBuf tap[5]; int tap_dims[5][4] = {};

  void save_tap(int which, const float* src, int n, int c, int h, int w, hipStream_t st) {
    if (!taps_on) return;
    const size_t bytes = (size_t)n * c * h * w * 4;
    tap[which].ensure(bytes);
    SS4K_HIP(hipMemcpyAsync(tap[which].ptr, src, bytes, hipMemcpyDeviceToDevice, st));
    tap_dims[which][0] = n; tap_dims[which][1] = c; tap_dims[which][2] = h; tap_dims[which][3] = w;
  }
  void out_shape(int h, int w, int* oh, int* ow) const {
    int lh = h, lw = w;
    if (cfg.single_mode) { lh = cfg.lr_h; lw = cfg.lr_w; }
    else if ((w > cfg.lr_w || h > cfg.lr_h) && cfg.lr_hr_resize) { lh = cfg.lr_h; lw = cfg.lr_w; }
    int oc, H, W; sr->out_shape(1, lh, lw, &oc, &H, &W);
    const bool resize = cfg.out_h > 0 && (cfg.single_mode || cfg.lr_hr_resize);
    *oh = resize ? cfg.out_h : H; *ow = resize ? cfg.out_w : W;
  }

  // diff = blur(hb) - blur(lb) (fsrcnn_upscaler.py:211-213), left in hb.  The blur is linear and its reflect padding commutes with
  // the subtraction, so ONE blur of hb - lb, in its separable form (two 17-tap passes instead of two 289-tap ones): the same
  // value up to the order of the fp32 additions (1e-7 relative; parity tolerance 1e-3 / 1e-4, colour tap vs oracle)
  void color_diff(int P, int mh, int mw, hipStream_t st) {
    op_sub(hb.as<float>(), lb.as<float>(), hbb.as<float>(), (size_t)P * mh * mw, st);
    op_gauss17_reflect(hbb.as<float>(), lbb.as<float>(), hb.as<float>(), k_gauss17.as<float>(), P, mh, mw, st);
  }

  // fsrcnn_upscaler.py:168-233
  void multi(const uint8_t* in, int n, int h, int w, uint8_t* out, hipStream_t st) {
    const int P = 3 * n;
    img.ensure((size_t)P * h * w * 4);
    op_u8nhwc_to_f32nchw(in, img.as<float>(), n, h, w, 3, st);
    const float* lrp = img.as<float>(); int lh = h, lw = w;
    if ((w > cfg.lr_w || h > cfg.lr_h) && cfg.lr_hr_resize) {
      lh = cfg.lr_h; lw = cfg.lr_w;
      lr.ensure((size_t)P * lh * lw * 4);
      op_area(img.as<float>(), lr.as<float>(), P, h, w, lh, lw, st);
      lrp = lr.as<float>();
    }
    int oc, H, W; sr->out_shape(n, lh, lw, &oc, &H, &W);
    // fp16 HR tensor where the network's tail can write one (an fp16 SRVGG): the fused path below makes four passes over it (x4 on
    // 720p: 2880 x 5120 x 3 per frame), half the bytes each.  The fp32 parity path (taps) and fp32 models keep fp32.
    const bool hr16 = !taps_on && sr->can_half_out();
    hr.ensure((size_t)P * H * W * (hr16 ? 2 : 4));
    float* hrp = hr.as<float>();
    __half* hrh = hr.as<__half>();
    SS4K_REQUIRE(P <= STATS_MAX_PLANES, "too many frames in one job");
    st_hr.ensure(P * 8); st_lr.ensure(P * 8); st_acc.ensure(sizeof(double) * 2 * P * STATS_SLOTS);
    const int mh = H / 8, mw = W / 8;
    const bool color = mh > 8 && H > 64 && W > 64;  // local colour match, :201-218
    // the reference's guard looks at the height only; for HR widths of 65..71 its 17-tap reflect pad (8)
    // reaches the 8-pixel-wide map and torch raises - so does this build
    SS4K_REQUIRE(!color || mw > 8, "local colour match: HR width / 8 must exceed the 17-tap blur's reflect padding (torch raises here too)");
    const bool resize = cfg.out_h > 0 && cfg.lr_hr_resize && !(cfg.out_h == H && cfg.out_w == W);
    const double tm0 = now_ms();
    if (!taps_on) sr->out_stats_acc = st_acc.as<double>();   // statistics of hr ride along with its producer where it can
    sr->out_half = hr16;
    sr->forward(lrp, hrp, n, lh, lw, st);
    enq_model_ms = now_ms() - tm0; enq_denoise_ms = 0;
    sr->out_stats_acc = nullptr;
    if (!taps_on) {
      // ---- fused path: the HR tensor is written once by the network, then read by the statistics (unless they rode
      // along), by the area reduction and by ONE tail pass; every per-element expression is the unfused path's
      SS4K_REQUIRE(!hr16 || sr->out_stats_done, "internal: the fp16 HR tensor's statistics must ride along with its producer");
      if (sr->out_stats_done) op_plane_stats_finish(st_acc.as<double>(), st_hr.as<float>(), P, H * W, st);
      else op_plane_stats(st_acc.as<double>(), hrp, st_hr.as<float>(), P, H * W, st);
      op_plane_stats(st_acc.as<double>(), lrp, st_lr.as<float>(), P, lh * lw, st);
      const float* diff = nullptr;
      if (color) {
        const size_t sm = (size_t)P * mh * mw * 4;
        lb.ensure(sm); hb.ensure(sm); lbb.ensure(sm); hbb.ensure(sm);
        op_area(lrp, lb.as<float>(), P, lh, lw, mh, mw, st);
        if (hr16) op_area_normalized(hrh, hb.as<float>(), P, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
        else op_area_normalized(hrp, hb.as<float>(), P, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
        color_diff(P, mh, mw, st);
        diff = hb.as<float>();
      }
      if (!resize) {
        // normalise, - diff, clamp, * 255 -> uint8 NHWC in one read of hr
        if (hr16) op_tail_fused(hrh, out, diff, n, 3, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
        else op_tail_fused(hrp, out, diff, n, 3, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
      } else if (hr16) {
        op_tail_fused(hrh, static_cast<uint8_t*>(nullptr), diff, n, 3, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
        op_bicubic_u8(hrh, out, n, 3, H, W, cfg.out_h, cfg.out_w, st);
      } else {
        // normalise, - diff, clamp in place (bicubic reads 16 neighbours of the finished tensor), then bicubic -> uint8
        op_tail_fused(hrp, static_cast<uint8_t*>(nullptr), diff, n, 3, H, W, mh, mw, st_hr.as<float>(), st_lr.as<float>(), st);
        op_bicubic_u8(hrp, out, n, 3, H, W, cfg.out_h, cfg.out_w, st);
      }
      return;
    }
    // ---- unfused path (parity taps enabled): one kernel per torch call of the reference
    save_tap(0, lrp, n, 3, lh, lw, st); save_tap(1, hrp, n, 3, H, W, st);
    op_plane_stats(st_acc.as<double>(), hrp, st_hr.as<float>(), P, H * W, st);
    op_plane_stats(st_acc.as<double>(), lrp, st_lr.as<float>(), P, lh * lw, st);
    op_normalize(hrp, st_hr.as<float>(), st_lr.as<float>(), P, H * W, st);
    save_tap(2, hrp, n, 3, H, W, st);
    if (color) {
      const size_t sm = (size_t)P * mh * mw * 4;
      lb.ensure(sm); hb.ensure(sm); lbb.ensure(sm); hbb.ensure(sm);
      op_area(lrp, lb.as<float>(), P, lh, lw, mh, mw, st);
      op_area(hrp, hb.as<float>(), P, H, W, mh, mw, st);
      color_diff(P, mh, mw, st);
      op_bilinear(hb.as<float>(), hrp, P, mh, mw, H, W, /*subtract_from_out=*/1, 0, st);   // hr -= diff (:217)
    }
    save_tap(3, hrp, n, 3, H, W, st);
    op_clamp01(hrp, (size_t)P * H * W, st);
    const float* fin = hrp; int FH = H, FW = W;
    // always bicubic (quirk, :224-231).  At equal size align_corners=False bicubic has taps (0,1,0,0):
    // the identity on already clamped values, so that pass is skipped
    if (resize) {
      FH = cfg.out_h; FW = cfg.out_w;
      hr2.ensure((size_t)P * FH * FW * 4);
      op_bicubic(hrp, hr2.as<float>(), P, H, W, FH, FW, 1, st);
      fin = hr2.as<float>();
    }
    save_tap(4, fin, n, 3, FH, FW, st);
    op_f32nchw_to_u8nhwc(fin, out, n, 3, FH, FW, st);
  }

  // fsrcnn_upscaler.py:235-326.  The reference loops frame by frame in Python (:158-161); every
  // frame is independent (BSVD sees F = 1, only the noise-map level differs for the very first frame
  // of the stream), so the n frames of a job are pushed through each stage as one batch.
  void single(const uint8_t* in, int n, int h, int w, uint8_t* out, hipStream_t st) {
    const int lh = cfg.lr_h, lw = cfg.lr_w, P = 3 * n;
    const size_t plane = (size_t)lh * lw;
    // FSRCNN on frames that need neither the area resize nor the denoiser reads the uint8 frames ITSELF (fsrcnn.hip, U8IN: the same
    // (float)byte / 255.0f) and the low-resolution statistics come straight from the bytes: the fp32 colour planes are never written
    // (round 6: one launch and 44 MB + 44 MB of traffic per four 720p frames less).  The parity path (taps) keeps the planes.
    const bool u8_direct = !taps_on && !cfg.denoising && !cfg.sr_is_realesrgan && h == lh && w == lw && sr->can_u8_in();
    if (u8_direct) {
      int oc, H, W; sr->out_shape(1, lh, lw, &oc, &H, &W);
      const bool hr16 = sr->can_half_out();
      hr.ensure((size_t)P * H * W * 4 * 2);
      st_hr.ensure(P * 8); st_lr.ensure(P * 8);
      SS4K_REQUIRE(P <= STATS_MAX_PLANES, "too many frames in one job");
      // one set of accumulators for both tensors' statistics - the frames' [0, P) and the network output's [P, 2 P) - finished by ONE launch
      // that also zeroes what it has read: the accumulators (their own buffer, sized once for the largest job) are memset only when they
      // are new or when a job died between its first partial sum and its finishing launch (three launches of ~ 5 us less than two
      // op_plane_stats calls, in a 0.65 ms job)
      SS4K_REQUIRE(2 * P <= STATS_MAX_PLANES, "too many frames in one job");
      const size_t acc2_bytes = sizeof(double) * 2 * STATS_MAX_PLANES * STATS_SLOTS;
      if (st_acc2.bytes < acc2_bytes) { st_acc2.ensure(acc2_bytes); acc2_clean = false; }
      // (a job that is being CAPTURED into a graph by the caller runs later, any number of times, in whatever state an eager job in
      // between has left: it always carries the memset, and nothing it records changes what the buffer holds now)
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(st, &cap);
      const bool capturing = cap != hipStreamCaptureStatusNone, was_clean = acc2_clean;
      if (!acc2_clean || capturing) SS4K_HIP(hipMemsetAsync(st_acc2.as<double>(), 0, acc2_bytes, st));
      acc2_clean = false;
      op_plane_stats_u8nhwc_partial(st_acc2.as<double>(), in, n, lh * lw, 2 * P, 0, st);
      enq_denoise_ms = 0;
      const double tm0 = now_ms();
      sr->out_half = hr16; sr->in_u8 = true;
      sr->forward(reinterpret_cast<const float*>(in), hr.as<float>(), P, lh, lw, st);
      enq_model_ms = now_ms() - tm0;
      const bool rs_ = cfg.out_h > 0 && !(cfg.out_h == H && cfg.out_w == W);
      auto finish = [&](auto* hrt) {
        op_plane_stats_partial(st_acc2.as<double>(), hrt, P, H * W, 2 * P, P, st);
        op_plane_stats_finish2(st_acc2.as<double>(), st_lr.as<float>(), st_hr.as<float>(), P, lh * lw, H * W, true, st);
        acc2_clean = capturing ? was_clean : true;
        if (!rs_) op_tail_fused(hrt, out, static_cast<const float*>(nullptr), n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
        else {
          op_tail_fused(hrt, static_cast<uint8_t*>(nullptr), static_cast<const float*>(nullptr), n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
          op_bicubic_u8(hrt, out, n, 3, H, W, cfg.out_h, cfg.out_w, st);
        }
      };
      if (hr16) finish(hr.as<__half>()); else finish(hr.as<float>());
      return;
    }
    img.ensure((size_t)P * h * w * 4);
    op_u8nhwc_to_f32nchw(in, img.as<float>(), n, h, w, 3, st);
    // area resize, unconditional in this path (:239-241); at equal size adaptive average pooling is the identity: no copy
    const float* lr_before = img.as<float>();
    if (!(h == lh && w == lw)) {
      lr.ensure(plane * P * 4);
      op_area(img.as<float>(), lr.as<float>(), P, h, w, lh, lw, st);
      lr_before = lr.as<float>();
    }
    const float* lr_cur = lr_before;
    if (cfg.denoising) {
      lr4.ensure(plane * 4 * n * 4); den.ensure(plane * P * 4 * 2);
      for (int i = 0; i < n; ++i) {
        const float noise = first_frame ? 0.05f : (float)(0.1 * cfg.denoise_rate);  // :262, :269-271
        first_frame = false;
        float* dst = lr4.as<float>() + plane * 4 * i;
        SS4K_HIP(hipMemcpyAsync(dst, lr_before + plane * 3 * i, plane * 3 * 4, hipMemcpyDeviceToDevice, st));
        fill_plane(dst + plane * 3, plane, noise, st);  // constant noise-map plane
      }
      float* den0 = den.as<float>(); float* den1 = den0 + plane * P;
      const double t0 = now_ms();
      dn->forward(lr4.as<float>(), den0, n, lh, lw, st);
      enq_denoise_ms = now_ms() - t0;
      // clamp(sharpen(den)) * 0.8 + 0.2 * lr   (:279-281)
      op_depthwise_reflect(den0, den1, k_sharp.as<float>(), P, lh, lw, 3, 1, lr_before, 0.8f, (float)(1 - 0.8), st);
      lr_cur = den1;
    }
    save_tap(0, lr_cur, n, 3, lh, lw, st);
    int oc, H, W; sr->out_shape(1, lh, lw, &oc, &H, &W);
    hr.ensure((size_t)P * H * W * 4 * 2);
    float* hrp = hr.as<float>();
    // fp16 HR tensor where the network can write one and nothing but the fused tail reads it (no HR sharpening pass, no taps)
    const bool hr16 = !taps_on && !cfg.denoising && sr->can_half_out();
    const double tm0 = now_ms();
    sr->out_half = hr16;
    if (cfg.sr_is_realesrgan) sr->forward(lr_cur, hrp, n, lh, lw, st);
    else sr->forward(lr_cur, hrp, P, lh, lw, st);  // FSRCNN on the colour planes (:297)
    enq_model_ms = now_ms() - tm0;
    if (hr16) {
      __half* hrh = hr.as<__half>();
      st_hr.ensure(P * 8); st_lr.ensure(P * 8);
      SS4K_REQUIRE(P <= STATS_MAX_PLANES, "too many frames in one job");
      st_acc.ensure(sizeof(double) * 2 * P * STATS_SLOTS);
      op_plane_stats(st_acc.as<double>(), hrh, st_hr.as<float>(), P, H * W, st);
      op_plane_stats(st_acc.as<double>(), lr_before, st_lr.as<float>(), P, lh * lw, st);
      const bool rs_ = cfg.out_h > 0 && !(cfg.out_h == H && cfg.out_w == W);
      if (!rs_) op_tail_fused(hrh, out, static_cast<const float*>(nullptr), n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
      else {
        op_tail_fused(hrh, static_cast<uint8_t*>(nullptr), static_cast<const float*>(nullptr), n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
        op_bicubic_u8(hrh, out, n, 3, H, W, cfg.out_h, cfg.out_w, st);
      }
      return;
    }
    if (cfg.denoising) {
      float* hs = hrp + (size_t)P * H * W;
      op_depthwise_reflect(hrp, hs, k_sharp_hr.as<float>(), P, H, W, 3, 1, nullptr, 0, 0, st);  // :298-299
      hrp = hs;
    }
    save_tap(1, hrp, n, 3, H, W, st);
    st_hr.ensure(P * 8); st_lr.ensure(P * 8);
    SS4K_REQUIRE(P <= STATS_MAX_PLANES, "too many frames in one job");
    st_acc.ensure(sizeof(double) * 2 * P * STATS_SLOTS);
    op_plane_stats(st_acc.as<double>(), hrp, st_hr.as<float>(), P, H * W, st);
    op_plane_stats(st_acc.as<double>(), lr_before, st_lr.as<float>(), P, lh * lw, st);
    if (!taps_on) {
      // fused tail: normalise -> clamp -> [bicubic] -> uint8 without writing the normalised tensor (same expressions)
      const bool rs_ = cfg.out_h > 0 && !(cfg.out_h == H && cfg.out_w == W);
      if (!rs_) op_tail_fused(hrp, out, nullptr, n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
      else {
        op_tail_fused(hrp, static_cast<uint8_t*>(nullptr), nullptr, n, 3, H, W, 1, 1, st_hr.as<float>(), st_lr.as<float>(), st);
        op_bicubic_u8(hrp, out, n, 3, H, W, cfg.out_h, cfg.out_w, st);
      }
      return;
    }
    op_normalize(hrp, st_hr.as<float>(), st_lr.as<float>(), P, H * W, st);
    save_tap(2, hrp, n, 3, H, W, st);
    op_clamp01(hrp, (size_t)P * H * W, st);
    const float* fin = hrp; int FH = H, FW = W;
    if (cfg.out_h > 0 && !(cfg.out_h == H && cfg.out_w == W)) {  // equal size: identity, see multi()
      FH = cfg.out_h; FW = cfg.out_w;
      hr2.ensure((size_t)P * FH * FW * 4);
      op_bicubic(hrp, hr2.as<float>(), P, H, W, FH, FW, 1, st);
      fin = hr2.as<float>();
    }
    save_tap(4, fin, n, 3, FH, FW, st);
    op_f32nchw_to_u8nhwc(fin, out, n, 3, FH, FW, st);
  }

  static void fill_plane(float* p, size_t n, float v, hipStream_t st) {
    // hipMemsetD32Async writes a 32-bit pattern
    uint32_t bits; std::memcpy(&bits, &v, 4);
    SS4K_HIP(hipMemsetD32Async((hipDeviceptr_t)p, (int)bits, n, st));
  }
};

}  // namespace ss4k

struct ss4k_upscaler { ss4k::Upscaler u; };

using namespace ss4k;

extern "C" {

int ss4k_abi_version(void) { return SS4K_ABI_VERSION; }
const char* ss4k_last_error(void) { return g_err; }

int ss4k_ctx_create(int dev, ss4k_ctx** out) {
  return guard([&] {
    SS4K_REQUIRE(out, "ss4k_ctx_create: out is NULL");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
      throw Error(SS4K_ENODEV, "no HIP device available (libss4k_hip has no CPU fallback)");
    SS4K_REQUIRE(dev >= 0 && dev < count, "ss4k_ctx_create: bad device index");
    SS4K_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop; SS4K_HIP(hipGetDeviceProperties(&prop, dev));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
      throw Error(SS4K_ENODEV, std::string("libss4k_hip is built for gfx950 only; device is ") + prop.gcnArchName);
    auto c = std::make_unique<ss4k_ctx>();
    c->device = dev; c->num_cu = prop.multiProcessorCount;
    *out = c.release();
  });
}
void ss4k_ctx_destroy(ss4k_ctx* c) {
  if (!c) return;
  for (auto& kv : c->scratch) kv.second.release();
  for (auto& e : c->prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : c->prof_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : c->prof_sections) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (c->lane_stream_) (void)hipStreamDestroy(c->lane_stream_);
  for (auto ls : c->lane_parked) (void)hipStreamDestroy(ls);
  for (auto& kv : c->cv_area) if (kv.second.uploaded) (void)hipEventDestroy(kv.second.uploaded);
  if (c->fork_event) (void)hipEventDestroy(c->fork_event);
  if (c->done_event) (void)hipEventDestroy(c->done_event);
  delete c;
}
int ss4k_ctx_device(const ss4k_ctx* c) { return c ? c->device : -1; }

size_t ss4k_model_param_count(const ss4k_model_desc* d) { return d ? model_param_count(*d) : 0; }

int ss4k_model_create(ss4k_ctx* ctx, const ss4k_model_desc* d, const float* w, size_t n, ss4k_model** out) {
  return guard([&] {
    SS4K_REQUIRE(ctx && d && w && out, "ss4k_model_create: NULL argument");
    SS4K_HIP(hipSetDevice(ctx->device));
    auto m = std::make_unique<ss4k_model>();
    m->m.ctx = ctx; m->m.desc = *d;
    m->m.build(w, n);
    *out = m.release();
  });
}
void ss4k_model_destroy(ss4k_model* m) { delete m; }
int ss4k_model_out_shape(const ss4k_model* m, int n, int h, int w, int* oc, int* oh, int* ow) {
  return guard([&] { SS4K_REQUIRE(m && oc && oh && ow, "NULL argument"); m->m.out_shape(n, h, w, oc, oh, ow); });
}
int ss4k_model_in_channels(const ss4k_model* m) { return m ? m->m.in_channels() : SS4K_EINVAL; }
int ss4k_model_workspace_bytes(ss4k_model* m, int n, int h, int w, size_t* bytes) {
  return guard([&] {
    SS4K_REQUIRE(m && bytes, "ss4k_model_workspace_bytes: NULL argument");
    *bytes = m->m.workspace_bytes(n, h, w);
  });
}
int ss4k_model_forward(ss4k_model* m, const float* in, float* out, int n, int h, int w, void* stream) {
  return guard([&] {
    SS4K_REQUIRE(m && in && out, "ss4k_model_forward: NULL argument");
    m->m.forward(in, out, n, h, w, (hipStream_t)stream);
  });
}

int ss4k_model_check(ss4k_model* m, int wait) {
  return guard([&] {
    SS4K_REQUIRE(m, "ss4k_model_check: NULL argument");
    m->m.check_async_error(wait != 0);
  });
}

int ss4k_upscaler_create(ss4k_ctx* ctx, const ss4k_upscale_cfg* cfg, ss4k_model* sr, ss4k_model* dn, ss4k_upscaler** out) {
  return guard([&] {
    SS4K_REQUIRE(ctx && cfg && sr && out, "ss4k_upscaler_create: NULL argument");
    SS4K_REQUIRE(cfg->lr_h > 0 && cfg->lr_w > 0, "lr_shape must be positive");
    SS4K_REQUIRE(!cfg->denoising || dn, "denoising requested without a BSVD model");
    SS4K_REQUIRE(!cfg->denoising || cfg->single_mode, "the reference only denoises on the per-frame path (fsrcnn_upscaler.py:109,168-233)");
    SS4K_REQUIRE(cfg->single_mode || cfg->sr_is_realesrgan, "the batched path requires a 3-channel SR model (fsrcnn_upscaler.py:180-184)");
    SS4K_REQUIRE((sr->m.in_channels() == 3) == (cfg->sr_is_realesrgan != 0), "sr_is_realesrgan does not match the SR model kind");
    auto u = std::make_unique<ss4k_upscaler>();
    u->u.ctx = ctx; u->u.cfg = *cfg; u->u.sr = &sr->m; u->u.dn = dn ? &dn->m : nullptr;
    auto up = [&](DevBuf& b, const std::vector<float>& v) {
      b.ensure(v.size() * 4);
      SS4K_HIP(hipMemcpy(b.ptr, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    };
    up(u->u.k_gauss17, gaussian_taps_1d(17, 8.0f));
    up(u->u.k_sharp, sharpen_taps(0.00002));
    up(u->u.k_sharp_hr, sharpen_taps(0.00007));
    *out = u.release();
  });
}
void ss4k_upscaler_destroy(ss4k_upscaler* up) {
  if (!up) return;
  Upscaler& u = up->u;
  for (DevBuf* b : {&u.k_gauss17, &u.k_sharp, &u.k_sharp_hr, &u.img, &u.lr, &u.lr4, &u.den, &u.hr, &u.hr2, &u.lb, &u.hb,
                    &u.lbb, &u.hbb, &u.st_hr, &u.st_lr, &u.st_acc, &u.st_acc2})
    b->release();
  for (auto& t : u.tap) t.release();
  delete up;
}
int ss4k_upscaler_reset(ss4k_upscaler* up) { if (!up) return SS4K_EINVAL; up->u.first_frame = true; return SS4K_OK; }
int ss4k_upscaler_out_shape(const ss4k_upscaler* up, int n, int h, int w, int* oh, int* ow) {
  (void)n;
  return guard([&] { SS4K_REQUIRE(up && oh && ow, "NULL argument"); up->u.out_shape(h, w, oh, ow); });
}
int ss4k_upscale_frames(ss4k_upscaler* up, const uint8_t* in, int n, int h, int w, uint8_t* out, size_t cap, void* stream) {
  return guard([&] {
    SS4K_REQUIRE(up && in && out, "ss4k_upscale_frames: NULL argument");
    SS4K_REQUIRE(n > 0 && h > 0 && w > 0, "ss4k_upscale_frames: empty batch");
    int oh, ow; up->u.out_shape(h, w, &oh, &ow);
    const size_t per = (size_t)oh * ow * 3;
    SS4K_REQUIRE(cap >= per * n, "ss4k_upscale_frames: output buffer too small");
    hipStream_t st = (hipStream_t)stream;
    if (up->u.cfg.single_mode) {
      up->u.single(in, n, h, w, out, st);
    } else {
      up->u.multi(in, n, h, w, out, st);
    }
  });
}
int ss4k_upscaler_last_enqueue_ms(const ss4k_upscaler* up, double* denoise_ms, double* model_ms) {
  if (!up || !denoise_ms || !model_ms) return SS4K_EINVAL;
  *denoise_ms = up->u.enq_denoise_ms; *model_ms = up->u.enq_model_ms;
  return SS4K_OK;
}
int ss4k_upscaler_enable_taps(ss4k_upscaler* up, int en) { if (!up) return SS4K_EINVAL; up->u.taps_on = en != 0; return SS4K_OK; }
int ss4k_upscaler_read_tap(ss4k_upscaler* up, int which, float* out, size_t cap, int dims[4], void* stream) {
  return guard([&] {
    SS4K_REQUIRE(up && which >= 0 && which < 5 && dims, "bad tap request");
    const int* d = up->u.tap_dims[which];
    for (int i = 0; i < 4; ++i) dims[i] = d[i];
    const size_t nflt = (size_t)d[0] * d[1] * d[2] * d[3];
    SS4K_REQUIRE(nflt > 0, "tap not recorded (enable taps before ss4k_upscale_frames)");
    if (out) {
      SS4K_REQUIRE(cap >= nflt, "tap buffer too small");
      SS4K_HIP(hipMemcpyAsync(out, up->u.tap[which].ptr, nflt * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
  });
}

// ---- granular ops ---------------------------------------------------------------------------
int ss4k_op_u8nhwc_to_f32nchw(ss4k_ctx* c, const uint8_t* in, float* out, int n, int h, int w, int ch, void* s) {
  return guard([&] { SS4K_REQUIRE(c && in && out, "NULL argument"); op_u8nhwc_to_f32nchw(in, out, n, h, w, ch, (hipStream_t)s); SS4K_HIP(hipGetLastError()); });
}
int ss4k_op_area_resize(ss4k_ctx* c, const float* in, float* out, int p, int h, int w, int oh, int ow, void* s) {
  return guard([&] { SS4K_REQUIRE(c && in && out, "NULL argument"); op_area(in, out, p, h, w, oh, ow, (hipStream_t)s); SS4K_HIP(hipGetLastError()); });
}
int ss4k_op_bicubic_resize(ss4k_ctx* c, const float* in, float* out, int p, int h, int w, int oh, int ow, void* s) {
  return guard([&] { SS4K_REQUIRE(c && in && out, "NULL argument"); op_bicubic(in, out, p, h, w, oh, ow, 0, (hipStream_t)s); SS4K_HIP(hipGetLastError()); });
}
int ss4k_op_bilinear_resize(ss4k_ctx* c, const float* in, float* out, int p, int h, int w, int oh, int ow, void* s) {
  return guard([&] { SS4K_REQUIRE(c && in && out, "NULL argument"); op_bilinear(in, out, p, h, w, oh, ow, 0, 0, (hipStream_t)s); SS4K_HIP(hipGetLastError()); });
}
int ss4k_op_depthwise_reflect(ss4k_ctx* c, const float* in, float* out, int p, int h, int w, const float* k2d, int k, void* s) {
  return guard([&] {
    SS4K_REQUIRE(c && in && out && k2d, "NULL argument");
    SS4K_REQUIRE(k >= 1 && k <= 17 && (k & 1), "kernel size must be odd and <= 17");
    float* taps = c->buf("dw_taps", 17 * 17 * 4).as<float>();
    SS4K_HIP(hipMemcpyAsync(taps, k2d, (size_t)k * k * 4, hipMemcpyHostToDevice, (hipStream_t)s));
    op_depthwise_reflect(in, out, taps, p, h, w, k, 0, nullptr, 0, 0, (hipStream_t)s);
    SS4K_HIP(hipGetLastError());
  });
}
// ---- cv2.resize(..., INTER_AREA), shrinking by a non-integer factor (glue.hip: k_cv_area_u8; oracle/cv_area.py states the algorithm)
namespace {
int cv_round(double v) { return (int)std::nearbyint(v); }   // saturate_cast<int>(double): round half to even (default rounding mode)
void cv_check_factor(double f) {
  SS4K_REQUIRE(f > 0.0 && f < 1.0, "cv area resize: factors must shrink (0 < f < 1)");
  const double scale = 1.0 / f;
  SS4K_REQUIRE(std::fabs(scale - std::nearbyint(scale)) >= 2.220446049250313e-16, "cv area resize: 1 / f is an integer - OpenCV's fast path (other rounding) is not implemented");
}
struct CvEnt { int si; float a; };
// computeResizeAreaTab: the entries of every output cell [d * scale, (d + 1) * scale), in order; ofs[d] = first entry of cell d
void cv_area_tab(int ssize, int dsize, double scale, std::vector<CvEnt>& ent, std::vector<int>& ofs) {
  ent.clear(); ofs.assign(dsize + 1, 0);
  for (int d = 0; d < dsize; ++d) {
    ofs[d] = (int)ent.size();
    const double fs1 = d * scale, fs2 = fs1 + scale, cell = std::min(scale, ssize - fs1);
    int s1 = (int)std::ceil(fs1), s2 = (int)std::floor(fs2);
    s2 = std::min(s2, ssize - 1);
    s1 = std::min(s1, s2);
    if (s1 - fs1 > 1e-3) ent.push_back({s1 - 1, (float)((s1 - fs1) / cell)});
    for (int sx = s1; sx < s2; ++sx) ent.push_back({sx, float(1.0 / cell)});
    if (fs2 - s2 > 1e-3) ent.push_back({s2, (float)(std::min(std::min(fs2 - s2, 1.), cell) / cell)});
  }
  ofs[dsize] = (int)ent.size();
}
}  // namespace
int ss4k_op_cv_area_shape(int h, int w, double fx, double fy, int* oh, int* ow) {
  return guard([&] {
    SS4K_REQUIRE(oh && ow && h > 0 && w > 0, "ss4k_op_cv_area_shape: bad argument");
    cv_check_factor(fx); cv_check_factor(fy);
    *oh = cv_round(h * fy); *ow = cv_round(w * fx);
    SS4K_REQUIRE(*oh > 0 && *ow > 0, "cv area resize: empty output");
  });
}
int ss4k_op_cv_area_resize_u8(ss4k_ctx* c, const uint8_t* in, uint8_t* out, size_t out_capacity, int n, int h, int w, int ch, double fx, double fy, void* s) {
  return guard([&] {
    SS4K_REQUIRE(c && in && out, "NULL argument");
    SS4K_REQUIRE(n > 0 && h > 0 && w > 0 && ch >= 1 && ch <= 4, "cv area resize: n, h, w > 0 and 1 <= channels <= 4");
    cv_check_factor(fx); cv_check_factor(fy);
    SS4K_HIP(hipSetDevice(c->device));
    const auto key = std::make_tuple(h, w, fx, fy);
    auto it = c->cv_area.find(key);
    if (it == c->cv_area.end()) {
      if (c->cv_area.size() >= 64) {   // an image server sees arbitrary sizes: bounded; the tables may still be read by enqueued launches
        SS4K_HIP(hipDeviceSynchronize());
        for (auto& kv : c->cv_area) if (kv.second.uploaded) (void)hipEventDestroy(kv.second.uploaded);
        c->cv_area.clear();
      }
      ss4k_ctx::CvAreaTab t;
      t.oh = cv_round(h * fy); t.ow = cv_round(w * fx);
      SS4K_REQUIRE(t.oh > 0 && t.ow > 0, "cv area resize: empty output");
      std::vector<CvEnt> xe, ye; std::vector<int> xo, yo;
      cv_area_tab(w, t.ow, 1.0 / fx, xe, xo);
      cv_area_tab(h, t.oh, 1.0 / fy, ye, yo);
      auto put = [&](const void* p, size_t bytes) { const size_t at = (t.host.size() + 15) & ~size_t(15); t.host.resize(at + bytes); std::memcpy(t.host.data() + at, p, bytes); return at; };
      t.xe = put(xe.data(), xe.size() * sizeof(CvEnt)); t.xo = put(xo.data(), xo.size() * sizeof(int));
      t.ye = put(ye.data(), ye.size() * sizeof(CvEnt)); t.yo = put(yo.data(), yo.size() * sizeof(int));
      t.dev.ensure(t.host.size());
      // upload and event on the LOCAL table; only a complete entry enters the cache (a throw here leaves no half-built entry behind whose
      // NULL event every later call for this shape would wait on).  Moving the table keeps its host buffer's address: the copy stays valid.
      SS4K_HIP(hipMemcpyAsync(t.dev.ptr, t.host.data(), t.host.size(), hipMemcpyHostToDevice, (hipStream_t)s));
      SS4K_HIP(hipEventCreateWithFlags(&t.uploaded, hipEventDisableTiming));
      const hipError_t rec = hipEventRecord(t.uploaded, (hipStream_t)s);
      if (rec != hipSuccess) {
        (void)hipStreamSynchronize((hipStream_t)s);   // the copy above reads t.host: let it finish before the table dies
        (void)hipEventDestroy(t.uploaded);
        SS4K_HIP(rec);
      }
      it = c->cv_area.emplace(key, std::move(t)).first;
    }
    auto& e = it->second;
    SS4K_REQUIRE(out_capacity >= (size_t)n * e.oh * e.ow * ch, "cv area resize: output buffer too small (ss4k_op_cv_area_shape gives the size)");
    SS4K_HIP(hipStreamWaitEvent((hipStream_t)s, e.uploaded, 0));
    const char* d = e.dev.as<char>();
    op_cv_area_u8(in, out, d + e.xe, reinterpret_cast<const int*>(d + e.xo), d + e.ye, reinterpret_cast<const int*>(d + e.yo), n, h, w, ch, e.oh, e.ow, (hipStream_t)s);
  });
}
int ss4k_op_plane_stats(ss4k_ctx* c, const float* in, float* stats, int p, int hw, void* s) {
  return guard([&] {
    SS4K_REQUIRE(c && in && stats, "NULL argument");
    op_plane_stats(c->buf("stats_acc", sizeof(double) * 2 * STATS_MAX_PLANES * STATS_SLOTS).as<double>(), in, stats, p, hw, (hipStream_t)s);
  });
}
int ss4k_op_f32nchw_to_u8nhwc(ss4k_ctx* c, const float* in, uint8_t* out, int n, int ch, int h, int w, void* s) {
  return guard([&] { SS4K_REQUIRE(c && in && out, "NULL argument"); op_f32nchw_to_u8nhwc(in, out, n, ch, h, w, (hipStream_t)s); SS4K_HIP(hipGetLastError()); });
}

#ifdef SS4K_DEV
int ss4k_bench_conv(ss4k_ctx* c, int dtype, int cin0, int cin1, int cout, int n, int h, int w, int flags, int iters,
                    double* avg_us, void* stream) {
  return guard([&] {
    SS4K_REQUIRE(c && avg_us && iters > 0, "bad argument");
    *avg_us = bench_conv_layer(c, dtype, cin0, cin1, cout, n, h, w, flags, iters, (hipStream_t)stream);
  });
}
#endif  // SS4K_DEV

// ---- profiling hooks --------------------------------------------------------------------------
static void prof_collect(ss4k_ctx* c) {
  for (auto& e : c->prof_events) {
    SS4K_HIP(hipEventSynchronize(e.b));
    float ms = 0; SS4K_HIP(hipEventElapsedTime(&ms, e.a, e.b));
    const int k = e.kind >= 0 && e.kind < PROF_KINDS ? e.kind : 0;
    c->kind_ms[k] += ms; c->kind_flops[k] += e.flops; c->kind_launches[k] += 1;
    if (k == PROF_CONV) { c->prof_ms += ms; c->prof_flops += e.flops; c->prof_launches += 1; }
    if (e.family) { auto& f = c->prof_families[e.family]; f.launches += 1; f.ms += ms; f.flops += e.flops; }
    c->prof_pool.push_back(e);
  }
  c->prof_events.clear();
  for (auto& e : c->prof_sections) {
    SS4K_HIP(hipEventSynchronize(e.b));
    float ms = 0; SS4K_HIP(hipEventElapsedTime(&ms, e.a, e.b));
    c->prof_section_ms += ms;
    c->prof_pool.push_back(e);
  }
  c->prof_sections.clear();
}
int ss4k_stream_pair_check(ss4k_ctx* c, void* a, void* b, int* side_by_side) {
  return guard([&] {
    SS4K_REQUIRE(c && side_by_side, "ss4k_stream_pair_check: NULL argument");
    SS4K_REQUIRE(a != b, "ss4k_stream_pair_check: the same stream twice");
    SS4K_HIP(hipSetDevice(c->device));
    *side_by_side = stream_pair_ok((hipStream_t)a, (hipStream_t)b) ? 1 : 0;
  });
}
int ss4k_prof_enable(ss4k_ctx* c, int en) { if (!c) return SS4K_EINVAL; c->prof = en != 0; return SS4K_OK; }
int ss4k_prof_reset(ss4k_ctx* c) {
  return guard([&] { SS4K_REQUIRE(c, "NULL ctx"); prof_collect(c); c->prof_ms = 0; c->prof_section_ms = 0; c->prof_flops = 0; c->prof_launches = 0;
    c->prof_families.clear();
    for (int k = 0; k < PROF_KINDS; ++k) { c->kind_ms[k] = 0; c->kind_flops[k] = 0; c->kind_launches[k] = 0; } });
}
int ss4k_prof_read_family(ss4k_ctx* c, int index, char* name, size_t name_capacity, int64_t* launches, double* ms, double* flops) {
  return guard([&] {
    SS4K_REQUIRE(c && index >= 0 && name && name_capacity > 0, "ss4k_prof_read_family: bad argument");
    prof_collect(c);
    SS4K_REQUIRE((size_t)index < c->prof_families.size(), "ss4k_prof_read_family: index past the last family");
    auto it = c->prof_families.begin();
    std::advance(it, index);
    std::snprintf(name, name_capacity, "%s", it->first.c_str());
    if (launches) *launches = it->second.launches;
    if (ms) *ms = it->second.ms;
    if (flops) *flops = it->second.flops;
  });
}
int ss4k_prof_read(ss4k_ctx* c, int64_t* launches, double* ms, double* flops) {
  return guard([&] {
    SS4K_REQUIRE(c, "NULL ctx");
    prof_collect(c);
    if (launches) *launches = c->prof_launches;
    if (ms) *ms = c->prof_ms;
    if (flops) *flops = c->prof_flops;
  });
}
int ss4k_prof_read_kind(ss4k_ctx* c, int kind, int64_t* launches, double* ms, double* flops) {
  return guard([&] {
    SS4K_REQUIRE(c && kind >= 0 && kind < PROF_KINDS, "ss4k_prof_read_kind: bad argument");
    prof_collect(c);
    if (launches) *launches = c->kind_launches[kind];
    if (ms) *ms = c->kind_ms[kind];
    if (flops) *flops = c->kind_flops[kind];
  });
}
int ss4k_prof_read_section_ms(ss4k_ctx* c, double* section_ms) {
  return guard([&] {
    SS4K_REQUIRE(c && section_ms, "NULL argument");
    prof_collect(c);
    *section_ms = c->prof_section_ms;
  });
}

}  // extern "C"
