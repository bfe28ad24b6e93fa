/*
 * ss4k.h — C ABI of libss4k_hip.so: the MI355X (gfx950) replacement for the arithmetic on
 * sharkshark-4k's per-frame upscale path (reference: src/upscale/, Python only).
 *
 * The reference has no C interface; every entry point below cites the Python call it replaces
 * (paths relative to the reference repo).  Conventions:
 *   - return 0 on success, a negative SS4K_E* code on failure; ss4k_last_error() gives the text
 *     (thread-local).  Nothing throws across this boundary.
 *   - *_dev pointers are device (HIP) pointers BORROWED from the caller (PyTorch owns the I/O
 *     tensors); the library owns weights and workspaces.
 *   - work is enqueued on the caller's stream (hipStream_t passed as void*; NULL = default
 *     stream) with no hidden synchronisation, like the reference's eager torch calls.  One
 *     exception, once per (ctx, stream): the first multi-frame fp16 forward from a stream tests
 *     that the ctx's second launch chain really runs beside it (~ 3 ms, the host waits for the
 *     stream once; see ss4k_stream_pair_check).
 *   - one ss4k_ctx per device; a ctx and its children are not thread-safe (the reference's
 *     worker is a single-threaded process loop, base_service.py:33-60).
 *   - a ctx and everything created from it (models, upscalers) is SINGLE-STREAM: a model reuses its
 *     activation workspace and an upscaler its staging buffers on every call, and the granular
 *     ss4k_op_* calls share small per-context scratch.  Calls that touch the same ctx must be
 *     enqueued on one stream (or be serialised by the caller with events); two services that
 *     should overlap on one GPU take one ctx each.  Matches the reference (one worker process,
 *     default stream, fsrcnn_upscaler.py:118-166).
 *   - there is no CPU fallback: without a HIP device ss4k_ctx_create fails.
 */
#ifndef SS4K_H
#define SS4K_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS4K_ABI_VERSION 3   /* 3: the register-stationary kernel and the cross-layer chain left the product library (dev library only): flag bits 8
                                 (NO_RS), 64 (NO_CHAIN), 128 (CHAIN), 2048 (DENSE) and 16384 (CONV5_RS) are no longer accepted;
                                 ss4k_prof_read_family, ss4k_stream_pair_check and ss4k_op_cv_area_* were added.  2: ss4k_model_desc.reserved[0] became the validated `flags` word */

enum { SS4K_OK = 0, SS4K_EINVAL = -22, SS4K_ENOMEM = -12, SS4K_EHIP = -5, SS4K_ENODEV = -19 };

typedef struct ss4k_ctx ss4k_ctx;
typedef struct ss4k_model ss4k_model;
typedef struct ss4k_upscaler ss4k_upscaler;

/* network families on the path */
enum ss4k_model_kind {
  SS4K_FSRCNN = 1,  /* model/fsrcnn/model.py:6-62 */
  SS4K_RRDBNET = 2, /* basicsr RRDBNet, instantiated realesrgan/factory.py:113-125 */
  SS4K_SRVGG = 3,   /* SRVGGNetCompact, realesrgan/factory.py:18-82 */
  SS4K_BSVD = 4     /* BSVD driven with F = 1, bsvd/model.py:467-588, fsrcnn_upscaler.py:277 */
};

/* arithmetic/storage type of activations and weights inside the network.
 * F32: fp32 storage, exact-fp32 MFMA (parity gate, rtol 1e-3 / atol 1e-4 vs PyTorch CPU).
 * F16: fp16 storage, fp32 accumulate (what RealESRGANer(half=True) + TensorRT fp16 does,
 *      realesrgan/factory.py:168,206-230).  FSRCNN with F16 (fsrcnn/factory.py:47-69 builds a TensorRT fp16 engine):
 *      fp16 operands and intermediates, fp32 accumulation, input and output planes stay fp32; FSRCNN with F32 is
 *      fp32-grade (hi/lo-split fp16 MFMA or, SS4K_MODEL_FS_EXACT, exact-fp32 kernels). */
enum ss4k_dtype { SS4K_F32 = 0, SS4K_F16 = 1 };

typedef struct ss4k_model_desc {
  int32_t kind;        /* ss4k_model_kind */
  int32_t dtype;       /* ss4k_dtype */
  int32_t scale;       /* FSRCNN: deconv stride 2|4; RRDBNet: 1|2|4; SRVGG: 2|4; BSVD: 1 */
  int32_t num_feat;    /* RRDBNet/SRVGG trunk width (64); multiple of 16 */
  int32_t num_block;   /* RRDBNet: #RRDB (23|6); SRVGG: num_conv (32|16) */
  int32_t num_grow_ch; /* RRDBNet growth (32) */
  int32_t bsvd_chns[3];/* BSVD U-Net widths (32,64,128) */
  int32_t bsvd_mid_ch; /* 32 */
  int32_t bsvd_interm_ch; /* 30 */
  int32_t bsvd_stream; /* 0: every frame of a forward call is independent - how the service drives BSVD
                          (F = 1, fsrcnn_upscaler.py:277); 1: the n frames of one ss4k_model_forward call are ONE
                          stream run through the bidirectional buffers (BSVD.forward, bsvd/model.py:515-580) */
  int32_t flags;       /* SS4K_MODEL_* bits below; 0 = defaults.  None of them changes what is computed beyond what its
                          comment says: they choose between kernels / schedules that the tests hold bit-identical (or, for
                          FS_EXACT, inside the path's tolerance), and exist so that a caller - and the parity tests - can
                          pin a route without environment variables */
  int32_t reserved[3];
} ss4k_model_desc;

enum {
  /* --- what a deployment may want to choose ---------------------------------------------------------------------------------- */
  SS4K_MODEL_FS_EXACT = 1,      /* FSRCNN: exact-fp32 kernels for every stage instead of the fp16 hi/lo-split matrix-core
                                   stages (fp32-grade, ~1e-6 of the exact ones); also chosen automatically when the
                                   checkpoint's range does not fit the split (a weight >= 6e4: Model::build, csrc/models.cpp) */
  SS4K_MODEL_ONE_CHAIN = 2,     /* a multi-frame job never runs as two concurrent launch chains (frame lanes) */
  SS4K_MODEL_TWO_CHAINS = 4,    /* ... always does (default: measured per shape over the first calls) */
  SS4K_MODEL_HR_F32 = 512,      /* fp16 SRVGG / fp16-mode FSRCNN on the service paths: keep the network's output tensor (x4 on 720p: 2880 x 5120 x 3
                                   per frame) in fp32; default: fp16 (half the bytes of the service's four passes over it; the uint8 frames
                                   differ by at most 1 LSB in a few per cent of the bytes - an fp16 model's own error is 30 dB above that) */
  SS4K_MODEL_NO_UPS_PRESUM = 8192, /* RRDBNet fp16: conv_up1 / conv_up2 (3x3 convs on a nearest-x2 up-sampled tensor) in the direct form.
                                   Default: two of the three input rows an output row reads are the same low-resolution row, so their two
                                   MFMAs per tap column run as one with the weight fragments added in fp16 (6 instead of 9 MFMAs per
                                   pixel): one more fp16 rounding of a weight sum, NOT bit-identical to the direct form */
  SS4K_MODEL_NO_W16 = 32768,    /* fp16 layers with 64-cout groups and an even number of 16-channel input planes (SRVGG body, RRDBNet conv5 / trunk /
                                   tail, BSVD) on the v_mfma_f32_32x32x16_f16 build of the tile (conv_dense.hip's wide kernel) instead of
                                   conv_w16.hip's v_mfma_f32_16x16x32_f16 build (default: the chip runs these layers at its power cap and holds
                                   a 10 % higher clock on that shape; SRVGG x4 720p + 12 %).  The two builds add the same products in a
                                   different order: results differ in the last bits, the accuracy against the oracle is the same */
  /* --- pins of a fallback route the library takes by itself for some shapes: BIT-IDENTICAL results, used by the parity tests -------- */
  SS4K_MODEL_TILE_ROWS_16 = 16, /* 32-cout layers (conv_mfma.hip) on 16-row tiles ... */
  SS4K_MODEL_TILE_ROWS_20 = 32, /* ... or on 20-row tiles (default: by image height) */
  SS4K_MODEL_NO_PAIR = 256,     /* BSVD: the full-resolution layer pairs (inc, outc) as two launches each instead of the fused
                                   row-marching kernel (conv_pair.hip) */
  SS4K_MODEL_NO_DENSE = 1024,   /* RRDBNet: conv1..conv4 of every dense block as four launches, never as two fused layer pairs
                                   (csrc/conv_dense.hip: (conv1, conv2) and (conv3, conv4) stream their shared input planes once and
                                   hand x1 / x3 over in LDS) - what a job whose planes exceed 4 GB gets anyway */
  SS4K_MODEL_NO_WIDE = 4096,    /* fp16 layers with 64-cout groups and a plain epilogue on conv_mfma.hip's <2,4,4> build - what a layer of a job
                                   whose planes exceed 4 GB gets anyway - instead of conv_dense.hip's single-layer build (bit-identical to it);
                                   implies NO_W16 */
  SS4K_MODEL_FLAGS_ALL = 1 | 2 | 4 | 16 | 32 | 256 | 512 | 1024 | 4096 | 8192 | 32768
};

int ss4k_abi_version(void);
const char* ss4k_last_error(void);

/* one per HIP device; replaces `.to(device)` / torch's implicit CUDA context. */
int ss4k_ctx_create(int hip_device, ss4k_ctx** out);
void ss4k_ctx_destroy(ss4k_ctx* ctx);
int ss4k_ctx_device(const ss4k_ctx* ctx);

/* Number of fp32 scalars ss4k_model_create expects: the model's state_dict tensors flattened and
 * concatenated in state_dict order (OIHW conv weights, then bias, PReLU slopes; ConvTranspose
 * weight as (C_in, C_out, kH, kW)).  Host-only, no GPU needed.  Returns 0 for a bad desc. */
size_t ss4k_model_param_count(const ss4k_model_desc* desc);

/* Replaces build_model(...) (fsrcnn/factory.py:5, realesrgan/factory.py:108, bsvd/factory.py:21):
 * repacks the state_dict blob into MFMA-fragment order and uploads it.  host_weights may be
 * freed on return. */
int ss4k_model_create(ss4k_ctx* ctx, const ss4k_model_desc* desc, const float* host_weights,
                      size_t n_floats, ss4k_model** out);
void ss4k_model_destroy(ss4k_model* m);

/* Output geometry of ss4k_model_forward for an (n, c, h, w) input. */
int ss4k_model_out_shape(const ss4k_model* m, int n, int h, int w, int* out_c, int* out_h, int* out_w);
/* Input channel count the model expects (FSRCNN 1, RRDBNet/SRVGG 3, BSVD 4). */
int ss4k_model_in_channels(const ss4k_model* m);

/* Device bytes of activation workspace the model holds after a forward of n frames of h x w (it grows
 * to the largest shape seen and is reused); nothing is allocated or launched by this call.  The
 * size query of SURVEY.md 8(b) (`ss4k_workspace_bytes`). */
int ss4k_model_workspace_bytes(ss4k_model* m, int n, int h, int w, size_t* bytes);
/* Replaces `self.model(x)` (fsrcnn_upscaler.py:181,293-297) and `self.denoise_model(x)`
 * (:277): x is contiguous NCHW fp32 in [0,1] on the device; result contiguous NCHW fp32.
 * BSVD: in (n,4,h,w) = the reference's (n,1,4,h,w); out (n,3,h,w).  FSRCNN: (planes,1,h,w). */
int ss4k_model_forward(ss4k_model* m, const float* in_nchw_dev, float* out_nchw_dev, int n, int h,
                       int w, void* hip_stream);

/* Asynchronous status of the model's earlier forwards.  No kernel of the product library has an asynchronous failure mode: always
 * SS4K_OK, at once.  (The dev library's cross-layer chain kernel - include/ss4k_dev.h, SS4K_DEV_MODEL_CHAIN - marks a sticky word when a
 * work unit gives up waiting for its neighbours; there this call returns SS4K_EHIP once per failure, after blocking until the model's
 * last chain launch has finished when wait != 0.) */
int ss4k_model_check(ss4k_model* m, int wait);

/* ---- the service's frame-in/frame-out hot path ------------------------------------------- */
typedef struct ss4k_upscale_cfg {
  int32_t lr_h, lr_w;       /* self.lr_shape (fsrcnn_upscaler.py:93-100) */
  int32_t out_h, out_w;     /* self.output_shape, 0,0 = None (fsrcnn_upscaler.py:107) */
  int32_t lr_hr_resize;     /* :92 */
  int32_t single_mode;      /* :109  (per-frame path upscale_single vs batched upscale_multi) */
  int32_t sr_is_realesrgan; /* upscaler_model == 'realesrgan' (model gets (1,3,H,W)); else FSRCNN planes */
  int32_t denoising;        /* :111 */
  int32_t reserved0;
  double denoise_rate;      /* :102 (python float = double) */
  int32_t reserved[6];
} ss4k_upscale_cfg;

/* Replaces FsrcnnUpscalerService.proc_init (fsrcnn_upscaler.py:118-139): binds the SR model,
 * the optional BSVD model, and builds the four fixed depthwise kernels (:135-138). */
int ss4k_upscaler_create(ss4k_ctx* ctx, const ss4k_upscale_cfg* cfg, ss4k_model* sr,
                         ss4k_model* denoise /* NULL unless cfg->denoising */, ss4k_upscaler** out);
void ss4k_upscaler_destroy(ss4k_upscaler* up);
/* forget `lr_prev` state: the next frame is a "first frame" again (noise map 0.05, :269-271) */
int ss4k_upscaler_reset(ss4k_upscaler* up);
int ss4k_upscaler_out_shape(const ss4k_upscaler* up, int n, int h, int w, int* out_h, int* out_w);
/* Host milliseconds the last ss4k_upscale_frames spent ENQUEUEING the denoise stage and the SR model:
 * exactly what the reference's 'fsrcnn.denoise' / 'fsrcnn.model' profiler spans measure (host
 * time.time() around asynchronous launches, fsrcnn_upscaler.py:276-278,290-300; util/profiler.py:12-24).
 * denoise_ms = 0 when the job did not denoise. */
int ss4k_upscaler_last_enqueue_ms(const ss4k_upscaler* up, double* denoise_ms, double* model_ms);

/* Replaces FsrcnnUpscalerService.upscale(frames) (fsrcnn_upscaler.py:144-326): uint8 NHWC
 * (n,h,w,3) device frames in, uint8 NHWC (n,out_h,out_w,3) device frames out. */
int ss4k_upscale_frames(ss4k_upscaler* up, const uint8_t* in_nhwc_dev, int n, int h, int w,
                        uint8_t* out_nhwc_dev, size_t out_capacity_bytes, void* hip_stream);

/* Parity taps: fp32 NCHW copies of the intermediates of the LAST ss4k_upscale_frames call
 * (the oracle exposes the same points).  which: 0 = lr (after pre-resize/denoise),
 * 1 = model output (after sharpen_hr if denoising), 2 = after mean/std match,
 * 3 = after local colour match (multi mode only), 4 = final float before *255 truncation.
 * Enable with ss4k_upscaler_enable_taps before the call; shape returned via dims[4] (n,c,h,w). */
int ss4k_upscaler_enable_taps(ss4k_upscaler* up, int enable);
int ss4k_upscaler_read_tap(ss4k_upscaler* up, int which, float* out_dev, size_t capacity_floats,
                           int dims[4], void* hip_stream);

/* ---- granular glue ops (each replaces one torch call on the path; used by the parity tests) */
/* img.permute(0,3,1,2) / 255.0  (fsrcnn_upscaler.py:170-171) */
int ss4k_op_u8nhwc_to_f32nchw(ss4k_ctx* ctx, const uint8_t* in_dev, float* out_dev, int n, int h, int w,
                              int c, void* hip_stream);
/* F.interpolate(mode='area') == adaptive average pooling (:174-176, :205-210, :239-241) */
int ss4k_op_area_resize(ss4k_ctx* ctx, const float* in_dev, float* out_dev, int planes, int h, int w,
                        int oh, int ow, void* hip_stream);
/* F.interpolate(mode='bicubic'), A=-0.75, align_corners=False (:226-228, :319-321) */
int ss4k_op_bicubic_resize(ss4k_ctx* ctx, const float* in_dev, float* out_dev, int planes, int h, int w,
                           int oh, int ow, void* hip_stream);
/* F.interpolate(mode='bilinear'), align_corners=False (:214-216) */
int ss4k_op_bilinear_resize(ss4k_ctx* ctx, const float* in_dev, float* out_dev, int planes, int h, int w,
                            int oh, int ow, void* hip_stream);
/* depthwise KxK conv with padding_mode='reflect' (blur_ker / sharpen_ker, :20-84); k2d_host = K*K
 * fp32 taps on the host, K = 3 or 17 (the service's sharpen and blur kernels) */
int ss4k_op_depthwise_reflect(ss4k_ctx* ctx, const float* in_dev, float* out_dev, int planes, int h, int w,
                              const float* k2d_host, int k, void* hip_stream);
/* per-plane mean and unbiased std (:192-197): stats_dev[2*p] = mean, [2*p+1] = std */
int ss4k_op_plane_stats(ss4k_ctx* ctx, const float* in_dev, float* stats_dev, int planes, int hw,
                        void* hip_stream);
/* (clamp(x,0,1) * 255).permute(0,2,3,1).to(uint8) — truncation (:232-233) */
int ss4k_op_f32nchw_to_u8nhwc(ss4k_ctx* ctx, const float* in_dev, uint8_t* out_dev, int n, int c, int h,
                              int w, void* hip_stream);

/* cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_AREA) on uint8 NHWC frames for SHRINKING factors whose inverse is not an integer -
 * the image server's pre / post scale (0.8 / 0.85 / 0.66: image_pipeline.py:149-150, 259-261, 272-273, 347-348), so that a caller can keep both
 * on the device next to the upload / download.  OpenCV's general area path (computeResizeAreaTab + ResizeArea_Invoker<uchar, float>), restated
 * in oracle/cv_area.py; PARITY UNPINNED: the reference pins no OpenCV version and cv2 is not in the build image.  Output size: (cvRound(h * fy),
 * cvRound(w * fx)) - ss4k_op_cv_area_shape (host only).  Other factors return SS4K_EINVAL.  The first call for a new (h, w, fx, fy) builds and
 * uploads two small tables (kept per context, at most 64 shapes; no synchronisation unless that cache overflows). */
int ss4k_op_cv_area_shape(int h, int w, double fx, double fy, int* out_h, int* out_w);
int ss4k_op_cv_area_resize_u8(ss4k_ctx* ctx, const uint8_t* in_nhwc_dev, uint8_t* out_nhwc_dev, size_t out_capacity_bytes, int n, int h, int w,
                              int channels, double fx, double fy, void* hip_stream);

/* ---- measurement hooks (bench.py: live per-kernel timing with HIP events on the launch stream) */
/* When enabled, every launch of the dominant conv kernel is bracketed with hipEvents on the
 * stream it is launched on; ss4k_prof_read returns (#launches, total ms, algorithmic FLOPs). */
int ss4k_prof_enable(ss4k_ctx* ctx, int enable);
int ss4k_prof_reset(ss4k_ctx* ctx);
int ss4k_prof_read(ss4k_ctx* ctx, int64_t* launches, double* total_ms, double* flops);
/* The same per kernel family: kind 0 = the 3x3 conv kernels (what ss4k_prof_read returns), 1 / 2 / 3 = FSRCNN's head (5x5 conv +
 * shrink), mapping (4 x conv3x3 12->12) and tail (expand + 9x9 transposed conv) stages, each bracketed as one unit. */
int ss4k_prof_read_kind(ss4k_ctx* ctx, int kind, int64_t* launches, double* total_ms, double* flops);
/* ... and per kernel BUILD of the conv launches (which tile / MFMA shape / epilogue form a launch was routed to), since the last reset:
 * index 0 .. n-1 in name order, SS4K_EINVAL past the last one.  `name` receives a NUL-terminated description that starts with the kernel's
 * C++ name as rocprofv3 prints it (e.g. "w16::conv3x3_w16_kernel<RL> (...)").  With two launch chains in flight the per-launch times overlap:
 * read these from a one-chain run (SS4K_MODEL_ONE_CHAIN) when a kernel's own rate is wanted. */
int ss4k_prof_read_family(ss4k_ctx* ctx, int index, char* name, size_t name_capacity, int64_t* launches, double* total_ms, double* flops);
/* ---- streams
 * HIP serves a process's streams from a few hardware queues.  Two streams on one queue run in order whatever the program says, and
 * some pairs of queues launch slowly while both are busy (measured: 14 us per launch instead of 2.5, profiles/earlier/r05/r05_lane_queue.txt);
 * which queue a stream gets depends on how many the process created before it.  This call MEASURES a pair (a 0.2 ms idle kernel on
 * each, then 200 x 1 us kernels interleaved; ~ 3 ms, synchronises both streams) and sets *side_by_side to 1 or 0.  A host that runs
 * several contexts on its own streams calls it once per pair and replaces a stream that fails (keep the failed one alive until the
 * replacement exists, or the new stream lands on the same queue).  The library does this itself for each ctx's internal lane stream
 * (once per ctx and caller stream, before the first two-chain forward); that built-in test is switched off by the environment variable
 * SS4K_NO_LANE_CHECK=1, is skipped while the caller's stream is being captured into a graph or a profiler has preloaded itself
 * (ROCP_TOOL_LIBRARIES / LD_PRELOAD of rocprofiler: counter collection serialises kernels), and leaves the NULL stream alone while any
 * other stream of the process is under a global-mode capture. */
int ss4k_stream_pair_check(ss4k_ctx* ctx, void* hip_stream_a, void* hip_stream_b, int* side_by_side);

/* Conv sections: wall time, on the caller's stream, from the first conv launch of every network forward to the end
 * of its last one (launch boundaries included).  A job's frames may go through the conv layers as two CONCURRENT launch
 * chains (frame lanes): the per-launch times of ss4k_prof_read then overlap, and sum(FLOPs) / section time is the rate
 * the chip sustained; total_ms / section_ms = average number of conv launches in flight. */
int ss4k_prof_read_section_ms(ss4k_ctx* ctx, double* section_ms);

#ifdef __cplusplus
}
#endif
#endif /* SS4K_H */
