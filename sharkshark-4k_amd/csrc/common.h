// Internal declarations shared by the translation units of libss4k_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <set>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/ss4k.h"
#ifdef SS4K_DEV
#include "../../include/ss4k_dev.h"
#endif

namespace ss4k {

void set_error(const char* fmt, ...);

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SS4K_HIP(expr)                                                                      \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess)                                                                   \
      throw ::ss4k::Error(SS4K_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)

#define SS4K_REQUIRE(cond, msg)                                        \
  do {                                                                 \
    if (!(cond)) throw ::ss4k::Error(SS4K_EINVAL, std::string(msg));   \
  } while (0)

// A device allocation owned by the object that holds it (a context's scratch, a model's layers and activations) and freed with it;
// movable, never copied; grows on demand, never shrinks (no hipMalloc in steady state: shapes repeat frame after frame).
struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; o.bytes = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); ptr = o.ptr; bytes = o.bytes; o.ptr = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void ensure(size_t need) {
    if (need <= bytes) return;
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
    size_t want = (need + 255) & ~size_t(255);
    SS4K_HIP(hipMalloc(&ptr, want));
    bytes = want;
  }
  void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; bytes = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

struct ProfEvent { hipEvent_t a, b; double flops; int kind; const char* family; };   // kind: PROF_* below; family: the launcher's static name for the kernel build it chose (or null)
struct ProfFamily { int64_t launches = 0; double ms = 0, flops = 0; };
enum { PROF_CONV = 0, PROF_FS_HEAD = 1, PROF_FS_MAP = 2, PROF_FS_TAIL = 3, PROF_KINDS = 4 };

}  // namespace ss4k

struct ss4k_ctx {
  int device = 0;
  int num_cu = 256;
  // named scratch for the granular ops
  std::map<std::string, ss4k::DevBuf> scratch;
  // conv-kernel profiling (bench.py roofline leg)
  bool prof = false;
  std::vector<ss4k::ProfEvent> prof_events;
  std::vector<ss4k::ProfEvent> prof_pool;
  int64_t prof_launches = 0;
  const char* prof_family = nullptr;   // set by the innermost launcher while a ProfScope is open: which kernel build the launch went to
  std::map<std::string, ss4k::ProfFamily> prof_families;   // per kernel build since the last reset (ss4k_prof_read_family)
  std::set<const void*> lds_attr_set;  // kernels whose dynamic-LDS limit was raised on this device
  double prof_ms = 0, prof_flops = 0;
  double kind_ms[ss4k::PROF_KINDS] = {}, kind_flops[ss4k::PROF_KINDS] = {}; int64_t kind_launches[ss4k::PROF_KINDS] = {};
  // frame lanes (models.h): the second launch chain's stream and the fork / join events, shared by every model of the
  // context (models run one after the other on the caller's stream); created on first use
  hipStream_t lane_stream_ = nullptr;
  hipEvent_t fork_event = nullptr, done_event = nullptr;
  hipStream_t lane_stream() {
    if (!lane_stream_) {
#ifdef SS4K_DEV
      if (const char* e = std::getenv("SS4K_LANE_PRIO")) {   // A/B switch: priority of the lane stream (0 normal, -1 high, 1 low)
        if (hipStreamCreateWithPriority(&lane_stream_, hipStreamNonBlocking, std::atoi(e)) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipStreamCreateWithPriority failed");
        return lane_stream_;
      }
#endif
      if (hipStreamCreateWithFlags(&lane_stream_, hipStreamNonBlocking) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipStreamCreateWithFlags failed");
    }
    return lane_stream_;
  }
  static hipEvent_t untimed_event(hipEvent_t& e) {
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipEventCreateWithFlags failed");
    return e;
  }
  hipEvent_t lane_fork() { return untimed_event(fork_event); }
  hipEvent_t lane_done() { return untimed_event(done_event); }
  // A HIP stream is a software queue mapped onto one of a few hardware queues.  Two streams on ONE hardware queue run strictly one after
  // the other, and some PAIRS of hardware queues are slow when both are busy (measured, round 5: the 5th queue of a process against the
  // NULL stream's, 14 us per launch instead of 2.5 - a 4-frame RRDBNet job lost 6 %).  Which queue a new stream gets depends on how many
  // streams the process made before it, so before the first fork from a caller's stream the pair is TESTED (lane_check, models.cpp) and a
  // lane stream that fails is parked (kept alive, so that its successor lands on another queue) and replaced.  It is tested against the NULL stream as well.
  // Costs 3-6 ms and a host synchronisation per (context, caller stream), once.
  std::set<hipStream_t> lane_checked;
  std::vector<hipStream_t> lane_parked;
  int lane_replaced = 0;
  void lane_check(hipStream_t caller);
  // conv sections (bench): wall time on the caller's stream from a forward's first conv launch to the end of its last
  std::vector<ss4k::ProfEvent> prof_sections;
  double prof_section_ms = 0;
  ss4k::ProfEvent prof_get_events() {
    ss4k::ProfEvent pe{};
    if (!prof_pool.empty()) { pe = prof_pool.back(); prof_pool.pop_back(); pe.flops = 0; pe.kind = 0; pe.family = nullptr; return pe; }
    if (hipEventCreate(&pe.a) != hipSuccess || hipEventCreate(&pe.b) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipEventCreate failed");
    return pe;
  }
  // bracket one launch (or one stage) with events on its stream; no-ops unless profiling is enabled
  ss4k::ProfEvent prof_begin(hipStream_t st, int kind) {
    ss4k::ProfEvent pe{};
    if (!prof) return pe;
    pe = prof_get_events(); pe.kind = kind;
    if (hipEventRecord(pe.a, st) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipEventRecord failed");
    return pe;
  }
  void prof_end(ss4k::ProfEvent pe, hipStream_t st, double flops) {
    if (!pe.a) return;
    if (hipEventRecord(pe.b, st) != hipSuccess) throw ss4k::Error(SS4K_EHIP, "hipEventRecord failed");
    pe.flops = flops; pe.family = prof_family; prof_family = nullptr; prof_events.push_back(pe);
  }
  const char* zero_page() {
    auto& b = scratch["zero_page"];
    if (!b.ptr) { b.ensure(256); (void)hipMemset(b.ptr, 0, 256); }
    return b.as<char>();
  }
  // cv2-INTER_AREA tables (ss4k_op_cv_area_resize_u8): one entry per (h, w, fx, fy) seen; the host copy lives as long as the entry, so the
  // upload needs no synchronisation; `uploaded` orders later calls from OTHER streams after it
  struct CvAreaTab { std::vector<char> host; ss4k::DevBuf dev; hipEvent_t uploaded = nullptr; int oh = 0, ow = 0; size_t xe = 0, xo = 0, ye = 0, yo = 0; };
  std::map<std::tuple<int, int, double, double>, CvAreaTab> cv_area;
  ss4k::DevBuf& buf(const std::string& name, size_t bytes) {
    auto& b = scratch[name];
    b.ensure(bytes);
    return b;
  }
};

namespace ss4k {

// Bracket of one launch (or stage) with profiling events: prof_begin on construction, prof_end by done(); a launch that throws in
// between gives its event pair back to the pool instead of leaking it.  A no-op unless profiling is enabled.
struct ProfScope {
  ss4k_ctx* ctx; hipStream_t st; ProfEvent pe; bool open;
  ProfScope(ss4k_ctx* c, hipStream_t s, int kind) : ctx(c), st(s), pe(c->prof_begin(s, kind)), open(true) {}
  void done(double flops) { if (open) { open = false; ctx->prof_end(pe, st, flops); } }
  ~ProfScope() { if (open && pe.a) ctx->prof_pool.push_back(pe); }
  ProfScope(const ProfScope&) = delete; ProfScope& operator=(const ProfScope&) = delete;
};

enum Act { ACT_NONE = 0, ACT_LRELU = 1, ACT_PRELU = 2, ACT_RELU6 = 3 };
enum Epi {
  EPI_NHWC = 0,        // out[(n,y,x)*ocs + oco + v]
  EPI_NHWC_SUB2 = 1,   // stride-2 conv: keep even (y,x) -> out[(n,y/2,x/2)]
  EPI_NHWC_PS2 = 2,    // PixelShuffle(2): virtual cout = sub*C' + c' -> out[(n,2y+dy,2x+dx)*ocs + oco + c']
  EPI_NCHW_F32 = 3     // final fp32 planes out[((n*C+c)*H+y)*W+x], c < cout_real
};

// One 3x3 / pad 1 / stride 1 convolution over "planes" activations (see conv_mfma.hip) as an implicit GEMM.
struct ConvArgs {
  const char* in0; size_t in0_plane_bytes; int in0_plane0, nchunks0;  // segment 0: one plane per K-chunk
  const char* in1; size_t in1_plane_bytes; int in1_plane0, nchunks1;  // optional segment 1 (dense concat is free)
  const char* zero_page;                // >= 64 zero bytes: DMA source for the zero padding
  int N, H, W;                          // conv grid (== output grid before SUB2/PS2)
  int n0;                               // first frame of this launch inside the tensors (frame lanes: a job's frames split
                                        // over concurrent launches); N counts this launch's frames
  int job_n;                            // frames of the whole job these tensors hold (0: n0 + N).  Kernels that keep 32-bit byte offsets inside a
                                        // plane are chosen on THIS - the same kernel for every launch chain of a job, whatever its n0
  float grid_share;                     // size the persistent grid for this share of the chip's workgroup slots (0 = all)
  int mb_override;                      // 0: tile height of the 32-cout layers by image height; 4 / 5: rows per wave forced (A/B)
  int s3;                               // 32-cout layers without residuals on the three-stage kernel (conv_s3.hip)
  int wide_rl;                          // ... and a residual that is the layer's own input may go through the matrix core there (conv5 of an RDB)
  int wide;                             // 64-cout-group layers with a plain epilogue on conv_dense.hip's single-layer build (conv3x3_wide_kernel)
  int no_band;                          // dev experiment: tiles dealt round-robin over the workgroups instead of one contiguous band per XCD
  int ups2;                             // input is the nearest-x2 upsampling of an (H/2, W/2) tensor
  int ups_presum;                       // ... and the wide kernel may add the weight fragments of the two taps that read the same input row
  const void* wpk;                      // packed weights [group][chunk][dx,ks][dy][nb][lane][E]
  const void* wrs;                      // conv_rs.hip layout [group][cout group][chunk32][tap][cb][lane][8] or null
  int rs_wide;                          // wrs is packed for the eight-wave variant of the shape
  const void* w16;                      // conv_w16.hip layout [group of 64][chunk pair][phase][dy][16-cout block][lane][8] or null
  const float* bias;                    // [cout_pad] virtual order
  const float* prelu;                   // [cout_pad] or null
  int prelu_le1;                        // every PReLU slope of the layer is <= 1: t >= 0 ? t : t s == max(t, t s)
  int act; float slope;
  float alpha, gamma;                   // v = (act(acc+bias)*alpha + res1)*gamma + res2
  const char* res1; size_t r1_plane_bytes; int r1_plane0;
  const char* res2; size_t r2_plane_bytes; int r2_plane0;
  int bsvd_resid;                       // channels < 3: v = res1 - v (bsvd/model.py:436-442)
  int epi;
  char* out; size_t out_plane_bytes; int out_plane0;
  int cout_real, cout_pad;
  int tiles_x, tiles_y;
  int reverse;                          // walk the tiles back to front (placement only, never results)
  double flops;                         // algorithmic FLOPs of this layer (profiling only)
  int dbg;                              // selects an ablation build (ss4k_bench_conv only; 0 in production)
  unsigned long long* dbg_buf;          // DBG_STAMP: per-workgroup phase cycle counters
};
// bytes of the fp16 plane a launch's pixel offsets must reach: all frames of the job, not only this launch's (frame lanes: lane 1 starts at n0 = N / 2)
inline double conv_plane_span_f16(const ConvArgs& a) { return (double)std::max(a.n0 + a.N, a.job_n) * a.H * a.W * 32.0; }
enum { DBG_NO_STORE = 1, DBG_NO_MMA = 2, DBG_NO_TILE_DMA = 4, DBG_NO_W_DMA = 8, DBG_NO_EPILOGUE = 16, DBG_STAMP = 32 };  // | tile-shape id << 8 (ss4k_bench_conv)

// ---- cross-layer execution of a chain of plain 32-cout-wide convs as ONE persistent launch (conv_chain.hip) -------------
// A chain is a list of ITEMS; an item is one conv layer restricted to one 32-cout group (a 64-cout layer = two items), over all
// tiles.  Work units (item, tile) are handed out in order from a queue; a unit may read what earlier layers wrote on its 3 x 3
// tile neighbourhood as soon as those units have finished - per-tile counters replace the kernel boundary.
struct ChainItem {
  const char* in0; size_t in0_plane_bytes; int in0_plane0, nchunks0;
  const char* in1; size_t in1_plane_bytes; int in1_plane0, nchunks1;
  const char* wpk;                      // this group's packed weights (ConvArgs.wpk + group offset)
  const float* bias;                    // this group's 32 bias values (virtual cout order)
  int act; float slope, alpha, gamma;   // ACT_NONE / ACT_LRELU only
  const char* res1; size_t r1_plane_bytes; int r1_plane0;
  const char* res2; size_t r2_plane_bytes; int r2_plane0;
  char* out; size_t out_plane_bytes; int out_plane0;   // out_plane0: first plane of this GROUP
  int newest;                           // first K-chunk whose plane the previous layer wrote (0: all of them)
  unsigned need_old, need_new;          // units that must have finished on every tile of the 3 x 3 neighbourhood before chunk 0 /
                                        // before chunk `newest` is read (and before anything is written)
  unsigned pub_need;                    // > 0: this unit's counter add waits until its OWN tile's counter has reached pub_need (the
                                        // second cout group of a layer publishes after the first, so that "layer's first unit
                                        // done" can be read off the counter)
  int pad_[2];
};
struct ChainArgs {
  const ChainItem* items; int nitems;
  unsigned* ctl;                        // [0] queue head, [1] error word, [4 ...] per-tile counters of finished units; zeroed per launch
  unsigned* err_sticky;                 // device pointer of the model's sticky error word (pinned, host-mapped): OR-ed into when a unit
                                        // times out, never reset by a launch
  const char* zero_page;
  int N, n0, H, W, tiles_x, tiles_y;
  int grid;                             // > 0: workgroups to launch (default: one per workgroup slot)
  int abl;                              // dev library: timing-only ablation build (conv_chain.hip)
  unsigned spin_limit;                  // dev library: polls before a unit gives up (0 = SPIN_LIMIT); fault injection for the error path
};
// every unit of the chain: fp16, plain epilogue, cout group of 32.  rows_per_wave 4 or 5 (16- / 20-row tiles).
void launch_conv_chain(ss4k_ctx* ctx, const ChainArgs& a, int rows_per_wave, hipStream_t st);
size_t conv_chain_ctl_bytes(int ntiles);
int conv_chain_tiles(int N, int H, int W, int rows_per_wave, int* tiles_x, int* tiles_y);

// launchers (conv_mfma.hip)
void launch_conv3x3(ss4k_ctx* ctx, const ConvArgs& a, int dtype, hipStream_t st);
// conv_pair.hip: two chained full-resolution 3x3 layers (32 channels wide) as one row-marching launch
struct PairArgs {
  const char* in; size_t in_plane_bytes; int in_plane0, planes_a;   // conv A input: planes_a (1 | 2) 16-channel planes
  const char* wA; const float* biasA;                               // packed fragments (pack.cpp, nb = 1) and bias of conv A (32 couts, ReLU6)
  const char* wB; const float* biasB;                               // ... of conv B (2 K-chunks, 32 couts)
  const char* res; size_t res_plane_bytes; int res_plane0;          // epi 1 | 2: the denoiser's skip tensor (channels 0..2: skip - conv)
  const char* zero_page;                                            // >= 16 zero bytes: DMA source of the zero padding
  unsigned long long* dbg_buf;                                      // dev library, SS4K_PAIR_STAMP=1: per-wave phase cycle counters
  char* out; size_t out_plane_bytes; int out_plane0;
  int epi;                                                          // conv B: 0 ReLU6 -> planes, 1 residual -> planes, 2 residual -> NCHW fp32
  int cout_real;                                                    // epi 2: output channels
  int n0, N, H, W, bands;
  float grid_share;
  double flops;
};
bool conv3x3_pair_eligible(int planes_a, int cout_pad_a, int nchunks_b, int cout_pad_b);
void launch_conv3x3_pair(ss4k_ctx* ctx, const PairArgs& a, hipStream_t st);

// conv_dense.hip: conv_k and conv_{k+1} of a dense block (32 couts each, LeakyReLU, fp16) as one launch; conv_{k+1} reads conv_k's
// input planes (segments 0 / 1, nchunks0 + nchunks1 K-chunks) plus conv_k's output, which it takes from LDS
struct DenseArgs {
  const char* in0; size_t in0_plane_bytes; int in0_plane0, nchunks0;
  const char* in1; size_t in1_plane_bytes; int in1_plane0, nchunks1;
  const char* w1; const float* bias1;                   // conv_k: packed fragments (pack.cpp, nb = 1), 32 biases
  const char* w2; const float* bias2;                   // conv_{k+1}: nchunks0 + nchunks1 + 2 K-chunks
  const char* w16p;                                     // conv_d16.hip: both layers' fragments (pack.cpp, pack_dense_d16) or null
  float slope;                                          // LeakyReLU of both layers
  char* out1; size_t out1_plane_bytes; int out1_plane0; // x_k (two planes)
  char* out2; size_t out2_plane_bytes; int out2_plane0; // x_{k+1}
  const char* zero_page;
  int n0, N, H, W, tiles_x, tiles_y;
  float grid_share;
  int reverse;
  double flops;
  unsigned long long* dbg_buf;                          // dev library, SS4K_DENSE_STAMP=1: per-wave phase cycle counters
};
bool conv3x3_dense2_eligible(int nchunks_a, int cout_pad_a, int nchunks_b, int cout_pad_b);
// the same fused pair on v_mfma_f32_16x16x32_f16 (conv_d16.hip): needs DenseArgs.w16p
bool conv3x3_d16_eligible(int nchunks_a, int cout_pad_a, int nchunks_b, int cout_pad_b);
void launch_conv3x3_d16(ss4k_ctx* ctx, const DenseArgs& a, hipStream_t st);
// conv_dense.hip: one layer, 64 couts per workgroup, plain epilogue (bit-identical to conv_mfma.hip's <__half,2,4,4> build)
struct ConvArgs;
bool conv3x3_wide_eligible(const ConvArgs& a, int dtype);
void launch_conv3x3_wide(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st);
void launch_conv3x3_dense2(ss4k_ctx* ctx, const DenseArgs& a, hipStream_t st);
// the same single-layer tile on v_mfma_f32_16x16x32_f16 (conv_w16.hip): needs ConvArgs.w16
bool conv3x3_w16_eligible(const ConvArgs& a, int dtype);
void launch_conv3x3_w16(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st);
// ... and its one-block build for layers with <= 4 output channels handed over as NCHW fp32 (conv_w16n.hip: RRDBNet's conv_last)
bool conv3x3_w16n_eligible(const ConvArgs& a, int dtype);
void launch_conv3x3_w16n(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st);

int conv_cw(int dtype);  // channels per plane / K-chunk: 16
// three-stage-ring build of the 32-cout tile body (conv_s3.hip)
bool conv3x3_s3_eligible(const ConvArgs& a, int dtype);
void launch_conv3x3_s3(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st);
// register-stationary-weights kernel (conv_rs.hip): fp16, plain epilogue, selected layer shapes
bool rs_config(int nplanes, int cout_pad, bool wide, int* nch, int* rows, int* cb, int* cg);
void launch_conv3x3_rs(ss4k_ctx* ctx, const ConvArgs& a, hipStream_t st);
inline int conv_rec_bytes(int dtype) { return dtype == SS4K_F16 ? 32 : 64; }  // bytes of one pixel's record in a plane

// weight packing (pack.cpp) --------------------------------------------------------------
struct PackSpec {
  int dtype;
  int cout_real, cin_total;        // OIHW dims of the source tensor
  std::vector<int> cin_map;        // [nchunks * CW] logical cin of every channel slot the conv reads, -1 = none
  int nchunks0, nchunks1;          // how the chunks split over the two input segments
  int ps2;                         // virtual cout order [sub][c'] for PixelShuffle(2)
  int force_nb1 = 0;               // pack a 64-cout layer as two 32-cout groups (conv_chain.hip runs every layer on the 32-cout tile body)
};
struct PackedConv {
  std::vector<uint8_t> w;          // device-order bytes
  std::vector<float> bias, prelu;  // [cout_pad]
  int cout_pad, nb, groups;
};
PackedConv pack_conv3x3(const PackSpec& s, const float* w_oihw, const float* bias, const float* prelu);
// conv_rs.hip weight order for a layer shape <nch, rows, cb> (fp16 only); same virtual cout order / bias as pack_conv3x3
std::vector<uint8_t> pack_conv3x3_rs(const PackSpec& s, const float* w_oihw, int cout_pad, int nch, int cb, int cg);
// conv_w16.hip weight order (fp16, 64-cout groups, an even number of K-chunks); same virtual cout order / bias as pack_conv3x3
std::vector<uint8_t> pack_conv3x3_w16(const PackSpec& s, const float* w_oihw, int cout_pad);
std::vector<uint8_t> pack_conv3x3_w16n(const PackSpec& s, const float* w_oihw);   // one 16-cout block (conv_w16n.hip)
// conv_d16.hip weight order of a dense-block layer pair
std::vector<uint8_t> pack_dense_d16(const PackSpec& sa, const float* wa, const PackSpec& sb, const float* wb);
int virt_to_real_cout(const PackSpec& s, int v);

}  // namespace ss4k
