// Model object behind ss4k_model (C ABI in include/ss4k.h).
#pragma once
#include "common.h"
#include "glue.h"
#include <tuple>

namespace ss4k {

// activation view in the "planes" layout (conv_mfma.hip): base, bytes per plane (N*H*W*record), first plane
struct Tens { char* p; size_t plane_bytes; int plane0; };

struct ConvLayer {
  DevBuf w, bias, prelu;
  DevBuf wch;              // 64-cout fp16 body layers: the weights once more as two 32-cout groups (conv_chain.hip)
  DevBuf wrs;              // conv_rs.hip weight order (fp16 layers of a supported shape, else empty)
  bool rs_wide = false;    // ... packed for the eight-wave variant
  DevBuf w16p;             // conv_d16.hip: this layer and the next as a fused dense-block pair (first layer of a pair only)
  DevBuf w16;              // conv_w16.hip weight order (fp16 layers with 64-cout groups and an even number of K-chunks, else empty)
  bool has_prelu = false;
  bool w16p_is_second = false;   // this layer is conv_{k+1} of a packed pair (it cannot start another one)
  bool prelu_le1 = false;  // every slope <= 1 (the epilogue may use max(t, t s))
  int cout_real = 0, cout_pad = 0, cin_real = 0, nchunks0 = 0, nchunks1 = 0;
};

struct ConvOpts {
  int ups2 = 0;
  int act = ACT_NONE; float slope = 0.f;
  float alpha = 1.f, gamma = 1.f;
  const Tens* res1 = nullptr; const Tens* res2 = nullptr;
  int bsvd_resid = 0;
  int epi = EPI_NHWC;
  Tens out{nullptr, 0, 0};
};

struct ParamCursor;
size_t model_param_count(const ss4k_model_desc& d);

struct Model {
  ss4k_ctx* ctx = nullptr;
  ss4k_model_desc desc{};
  std::vector<ConvLayer> layers;
  std::vector<DevBuf> acts;
  DevBuf fs_blob; FsrcnnWeights fsw{};
  size_t weight_bytes = 0;
  bool flip_walk = true;   // consecutive conv launches walk their tiles in opposite directions (cache reuse)
  int launch_parity = 0;
  int sub_batch = 0;       // > 0: a job's frames go through the network this many at a time (working set vs Infinity Cache)
  // plan_only: forward() walks the network without touching the device and records the activation
  // bytes each buffer slot would need (ss4k_model_workspace_bytes)
  bool plan_only = false;
  std::vector<size_t> plan_bytes;
  size_t workspace_bytes(int n, int h, int w);
  // one-shot request of the caller (the service): accumulate sum / sum of squares per output plane into this buffer
  // while the output is written (networks whose tail can do it set out_stats_done; the batch must not be split)
  double* out_stats_acc = nullptr;
  bool out_stats_done = false;
  // one-shot request of the caller (like out_stats_acc): write the NCHW output tensor as fp16 instead of fp32.  Only honoured where
  // can_half_out() says so (an fp16 SRVGG: its PixelShuffle tail converts anyway; an fp16-mode FSRCNN: its tail stores from registers); the caller sized `out` accordingly
  bool out_half = false;
  bool hr_f32 = false;         // SS4K_MODEL_HR_F32
  // one-shot request of the caller: `in` of the next forward is the service's uint8 NHWC frame tensor (n = 3 x frames colour planes), not fp32
  // planes.  Only where can_u8_in() says so (an FSRCNN on the matrix-core head: it converts while it loads)
  bool in_u8 = false;
  // (both matrix-core modes: the fp32-grade head sits at 256 registers and the byte loader costs it four spilled registers, + 4 % on that
  //  stage - the job still gains 2 % from the passes it no longer makes: profiles/NOTES_r06.md)
  bool can_u8_in() const { return desc.kind == SS4K_FSRCNN && !fs_exact; }
  bool can_half_out() const {
    return (desc.kind == SS4K_SRVGG || (desc.kind == SS4K_FSRCNN && !fs_exact)) && desc.dtype == SS4K_F16 && !plan_only && !hr_f32;
  }
  // frame lanes: the frames of a job are independent, so a batch of two or more frames can go through the conv layers as TWO concurrent
  // launch chains - lane 0 on the caller's stream, lane 1 on the context's lane stream, each working on its own half of
  // the frames of the same tensors, every launch still sized for the whole chip.  The hardware dispatcher then fills any
  // CU one chain leaves free (launch boundary, prologue, the partly filled last round of tiles) with waiting workgroups
  // of the other chain.  Frames are bit-identical to the single-chain path.
  bool use_s3 = false;         // 32-cout layers on the three-stage-ring kernel (conv_s3.hip)
  int mb_override = 0;         // SS4K_MB: rows per wave of the 32-cout layers' tiles forced to 4 or 5 (A/B switch)
  bool fs_exact = false;       // SS4K_FS_EXACT=1: FSRCNN's exact-fp32 kernels instead of the fp16-split matrix-core ones (A/B switch)
  int lanes_mode = 0;          // 0: measured per shape (lanes_begin), 1: one chain, 2: two chains
  float lane_grid_share = 1.f; // grid of a lane's launch as a share of the chip's workgroup slots (measured: 1.0 is best)
  int cur_lanes = 1, cur_n = 0;
  bool forked = false;
  struct LaneTune { int calls = 0, decided = 0; hipEvent_t ev[4][2] = {}; float ms[2] = {}; };
  // per (n, h, w); an image server fed arbitrary sizes must not grow this without bound: past LANE_TUNE_MAX shapes a new
  // shape runs one chain and is not measured
  static constexpr size_t LANE_TUNE_MAX = 64;
  std::map<std::tuple<int, int, int>, LaneTune> lane_tune;
  int tune_step(std::map<std::tuple<int, int, int>, LaneTune>& tab, int n, int h, int w, hipStream_t st);

  hipEvent_t tune_timed = nullptr;
  ProfEvent section{}; bool section_open = false;   // bench: wall time of a forward's conv launches
  void lanes_begin(int n, int h, int w, hipStream_t st);
  void lanes_join(hipStream_t st, bool end_of_forward);
  // cross-layer chain (conv_chain.hip): the RRDB body of small fp16 jobs as ONE persistent launch.  While chain_rec is set,
  // conv() records work items instead of launching; chain_run() resolves the dependencies and launches the chain.
  int dense_mode = 0;          // RRDBNet: (conv1, conv2) and (conv3, conv4) of every RDB as one fused launch each (conv_dense.hip): 0 = default (fused),
                               // 1 = never (SS4K_MODEL_NO_DENSE)
  bool use_wide = true;        // 64-cout-group fp16 layers with a plain epilogue on conv_dense.hip's single-layer build (SS4K_MODEL_NO_WIDE: conv_mfma.hip's <2,4,4>)
  bool use_d16 = false;        // dev experiment (SS4K_D16=1, dev library): fused dense-block pairs on conv_d16.hip (v_mfma_f32_16x16x32_f16, 14-row tiles)
  const float* raw_w_prev = nullptr; PackSpec raw_s_prev{}; int raw_li_prev = -1;   // build(): the previous add_conv's source weights (pair packing)
  bool use_w16 = true;         // ... on conv_w16.hip (v_mfma_f32_16x16x32_f16) where the layer has an even number of K-chunks and no up-sampled input (SS4K_MODEL_NO_W16: never)
  int conv5_mode = 0;          // RDB conv5: 0 = the 64-cout tile with the residual through the matrix core; 1 (dev library, SS4K_DEV_MODEL_CONV5_RS) = conv_rs.hip
  bool wide_rl = false;        // conv5 of an RDB on the wide kernel with its residual through the matrix core (when it is not routed to conv_rs.hip)
  bool ups_presum = true;      // RRDBNet's conv_up1 / conv_up2 on the wide kernel's pre-summed form (6 instead of 9 MFMAs per pixel; SS4K_MODEL_NO_UPS_PRESUM)
  int dense_mask = 3;          // ... which pairs: bit 0 = (conv1, conv2), bit 1 = (conv3, conv4)
  bool use_pair = true;        // BSVD: inc / outc layer pairs as one fused launch each (conv_pair.hip); SS4K_MODEL_NO_PAIR: two launches
  int chain_mode = 1;          // 1: never; 2 (dev library, SS4K_DEV_MODEL_CHAIN): the RRDB body of every fp16 job as one persistent launch
  bool chain_rec = false;
  struct ChainLayerRec { int first_item, nitems; const char* out_lo; const char* out_hi; double flops; };
  std::vector<ChainItem> chain_items;
  std::vector<ChainLayerRec> chain_layers;
  std::vector<ChainItem> chain_uploaded;
  DevBuf chain_tab, chain_ctl;
  unsigned* chain_err_host = nullptr;   // sticky error word of the chain kernel: pinned host memory the device ORs into (never reset by a launch)
  unsigned* chain_err_dev = nullptr;    // ... its device address
  hipEvent_t chain_done = nullptr; bool chain_pending = false;   // end of the last chain launch (ss4k_model_check(wait))
  void check_async_error(bool wait);    // throws if a chain launch reported a timed-out unit since the last check; clears the word
  void chain_record(const ConvArgs& a, const ConvLayer& L);
  void chain_run(int N, int H, int W, hipStream_t st);
  int fail_at_conv = 0, conv_calls = 0, forward_calls = 0;   // dev library only: fault injection (SS4K_FAIL_AT_CONV)
  int dbg = 0;  // ablation build selector forwarded to the conv kernel (bench only)
  unsigned long long* dbg_buf = nullptr;

  void build(const float* w, size_t n);
  // forward() = forward_impl() inside a guard: if anything throws after the second launch chain was forked, the caller's
  // stream is made to wait for the lane stream (the next call must not race lane-1 kernels still reading the activation
  // buffers), and the one-shot requests (out_stats_acc, the open profiling section, the timed tuning call) are dropped
  void forward(const float* in, float* out, int n, int h, int w, hipStream_t st);
  void forward_impl(const float* in, float* out, int n, int h, int w, hipStream_t st);
  void abort_forward(hipStream_t st) noexcept;
  void out_shape(int n, int h, int w, int* oc, int* oh, int* ow) const;
  int in_channels() const;
  int rs_mask = 32;        // layer shapes routed to conv_rs.hip (bit per shape, models.cpp rs_shape_bit); default: RDB conv5
  bool rs_wide = false;    // eight-wave variants of the 32-cout RS shapes (SS4K_RS_W8=1: A/B switch)
  bool use_rs = false;     // dev library: pack conv_rs.hip weights for the shapes in rs_mask and route those layers to the register-stationary kernel
  bool no_rl = false;      // dev library (SS4K_NO_RL=1): conv5's residual read from memory in the epilogue instead of through the matrix core
  ~Model() {
    // (device buffers are DevBuf members: freed with the object)
    if (chain_err_host) (void)hipHostFree(chain_err_host);
    if (chain_done) (void)hipEventDestroy(chain_done);
    for (auto* tab : {&lane_tune}) for (auto& t : *tab) for (auto& pr : t.second.ev) for (auto e : pr) if (e) (void)hipEventDestroy(e);
  }

  int add_conv(ParamCursor& pc, int cout, int cin_total, PackSpec spec, bool has_prelu_after, bool allow_rs = false, bool chainable = false);
  PackSpec spec_plain(int cin_real, int ps2 = 0) const;
  PackSpec spec_concat(int c0, int c1) const;
  PackSpec spec_masked(int c) const;
  PackSpec spec_shifted(int c) const;
  // planes holding the time-shifted channels [0, c/4) of a BiBufferConv input (whole planes)
  int shifted_planes(int c) const { return std::min(planes_for(c), (c / 4 + cw() - 1) / cw()); }
  int cw() const { return conv_cw(desc.dtype); }
  int rec() const { return conv_rec_bytes(desc.dtype); }  // bytes per pixel record of a plane
  int planes_for(int channels) const { return (channels + cw() - 1) / cw(); }
  void conv(int li, const Tens& in0, const Tens* in1, int N, int H, int W, const ConvOpts& o, hipStream_t st);
  // layers li (ReLU6) and li + 1 (options o) as one fused launch; false (nothing done) if the pair does not fit the fused kernel
  bool conv_pair(int li, const Tens& in0, int N, int H, int W, const ConvOpts& o, hipStream_t st);
  // dense-block layers li and li + 1 (both LeakyReLU `slope`, outputs out1 / out2) as one fused launch; false if the pair does not fit
  bool conv_dense(int li, const Tens& in0, const Tens* in1, int N, int H, int W, float slope, const Tens& out1, const Tens& out2, hipStream_t st);
  Tens act(int idx, size_t pixels, int channels);
  void pack_in(const float* in, const Tens& dst, int nplanes, int n, int c, int h, int w, int r, hipStream_t st);
};

}  // namespace ss4k

namespace ss4k {
double bench_conv_layer(ss4k_ctx* ctx, int dtype, int cin0, int cin1, int cout, int n, int h, int w, int flags,
                        int iters, hipStream_t st);
}
struct ss4k_model { ss4k::Model m; };
