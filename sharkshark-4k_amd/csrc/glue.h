// Launchers of the glue kernels (glue.hip) and FSRCNN kernels (fsrcnn.hip).
#pragma once
#include "common.h"

namespace ss4k {

void op_u8nhwc_to_f32nchw(const uint8_t* in, float* out, int n, int h, int w, int c, hipStream_t st);
void op_area(const float* in, float* out, int planes, int h, int w, int oh, int ow, hipStream_t st);
// acc: 2 * planes doubles of scratch owned by the caller (each upscaler owns its own, so two upscalers
// of one context never share partial sums), planes <= STATS_MAX_PLANES
constexpr int STATS_MAX_PLANES = 4096;
// the fp64 partial sums of a plane are spread over this many slots ([slot][plane][2], acc buffers hold STATS_SLOTS * 2 * planes
// doubles): thousands of workgroups adding into ONE address pair per plane serialise in the memory-side atomic units
// (measured: the statistics pass ran at 1.4 TB/s, the PixelShuffle tail doubled its time when the sums rode along)
constexpr int STATS_SLOTS = 32;
template <typename HT>
void op_plane_stats(double* acc, const HT* in, float* stats, int planes, int hw, hipStream_t st);
void op_plane_stats_finish(const double* acc, float* stats, int planes, int hw, hipStream_t st);
// mean / unbiased std of the colour planes of uint8 NHWC frames as the fp32 planes (float)byte / 255.0f would give them (same accumulators)
void op_plane_stats_u8nhwc(double* acc, const uint8_t* in, float* stats, int n, int hw, hipStream_t st);
// the two halves of the above and of op_plane_stats, for a caller that accumulates two tensors' statistics (accumulators zeroed by one memset:
// acc holds 2 x planes plane records, [0, planes) and [planes, 2 planes)) and finishes both with one launch
void op_plane_stats_u8nhwc_partial(double* acc, const uint8_t* in, int n, int hw, int acc_planes, int plane0, hipStream_t st);
template <typename HT>
void op_plane_stats_partial(double* acc, const HT* in, int planes, int hw, int acc_planes, int plane0, hipStream_t st);
void op_plane_stats_finish2(double* acc, float* stats_a, float* stats_b, int planes, int hw_a, int hw_b, bool rezero, hipStream_t st);
// HT: element type of the network's HR output tensor (float, or __half where the network's tail can write it)
template <typename HT>
void op_area_normalized(const HT* in, float* out, int planes, int h, int w, int oh, int ow, const float* st_hr, const float* st_lr,
                        hipStream_t st);
// fused tails (see glue.hip): any of st_hr/st_lr (normalise), diff (local colour match), out_u8 (final frame) may be null
template <typename HT>
void op_tail_fused(HT* hr, uint8_t* out_u8, const float* diff, int n, int c, int h, int w, int dh, int dw, const float* st_hr,
                   const float* st_lr, hipStream_t st);
template <typename HT>
void op_bicubic_u8(const HT* in, uint8_t* out, int n, int c, int h, int w, int oh, int ow, hipStream_t st);
void op_normalize(float* x, const float* st_hr, const float* st_lr, int planes, int hw, hipStream_t st);
void op_depthwise_reflect(const float* in, float* out, const float* taps_dev, int planes, int h, int w, int k,
                          int clamp01, const float* blend_src, float blend_a, float blend_b, hipStream_t st);
// separable form of the service's 17x17 sigma-8 reflect-padded Gaussian: in -> (horizontal) tmp -> (vertical) out
void op_gauss17_reflect(const float* in, float* tmp, float* out, const float* g17_dev, int planes, int h, int w, hipStream_t st);
void op_bilinear(const float* in, float* out, int planes, int h, int w, int oh, int ow, int subtract_from_out,
                 int clamp01, hipStream_t st);
void op_bicubic(const float* in, float* out, int planes, int h, int w, int oh, int ow, int clamp01, hipStream_t st);
void op_sub(const float* a, const float* b, float* out, size_t n, hipStream_t st);
void op_clamp01(float* x, size_t n, hipStream_t st);
// cv2 INTER_AREA shrink of uint8 NHWC frames (glue.hip); xe / ye: {int source index, float weight} entries, xo / yo: first entry of each output index (+ end)
void op_cv_area_u8(const uint8_t* in, uint8_t* out, const void* xe, const int* xo, const void* ye, const int* yo, int n, int h, int w, int c, int oh, int ow,
                   hipStream_t st);
void op_lane_spin(unsigned ticks, hipStream_t st);   // occupies `st` for ticks / 100 MHz (at most 0.5 ms) with one idle wave
bool stream_pair_ok(hipStream_t a, hipStream_t b);     // models.cpp: do the two streams run side by side at full launch rate? (measured, ~ 3 ms, synchronises)
void op_f32nchw_to_u8nhwc(const float* in, uint8_t* out, int n, int c, int h, int w, hipStream_t st);
template <typename T>
void op_pack_input(const float* in, T* out, int n, int c, int h, int w, int r, int nplanes, hipStream_t st);
template <typename T, typename HT>
void op_ps_nchw_addbase(const T* src, HT* out, const float* base, int n, int h, int w, int r, int cq, double* stats_acc, hipStream_t st);

void op_temporal_shift(const void* in, void* out, int nplanes, int frames, size_t frame_px, int slots_per_record,
                       int ch_per_plane, int fold, hipStream_t st);

// FSRCNN (fsrcnn.hip): whole-network forward on fp32 planes, weights in the device layout
// produced by fsrcnn_pack_weights.
struct FsrcnnWeights {
  // all fp32 on device
  const float* w_feat;   // [25][56]   tap-major, cout-minor
  const float* b_feat;   // [56]
  const float* a_feat;   // [56] PReLU
  const float* w_shrink; // [56][12]
  const float* b_shrink; const float* a_shrink;  // [12]
  const float* w_map[4]; // [9][12][12]  tap, cin, cout
  const float* b_map[4]; const float* a_map[4];
  const float* w_expand; // [12][56]
  const float* b_expand; const float* a_expand;  // [56]
  const float* w_deconv; // [81][56]  (ky*9+kx), cin
  float b_deconv;
  bool prelu_abs = false;   // matrix-core modes: producing weights / biases scaled by (1 + s) / 2 and the slope arrays hold (1 - s) / (1 + s) (models.cpp): PReLU is y + c |y|
};
// mode: fp32 accuracy on the fp16 matrix rate (hi/lo-split operands, the default of an SS4K_F32 model), the exact-fp32 kernels, or
// plain fp16 operands with fp32 accumulation (an SS4K_F16 model: the precision the reference's TensorRT engine runs FSRCNN in)
enum { FS_MODE_SPLIT = 0, FS_MODE_EXACT = 1, FS_MODE_HALF = 2 };
void fsrcnn_forward(ss4k_ctx* ctx, const FsrcnnWeights& W, int factor, const float* in, float* out, int planes, int h,
                    int w, float* ws12a, float* ws12b, int mode, hipStream_t st, bool out_half = false,
                    bool in_u8 = false);   // out_half: fp16 mode only, HR planes as fp16; in_u8: `in` is the uint8 NHWC frame tensor (planes / 3 frames)

}  // namespace ss4k
