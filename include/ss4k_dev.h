/*
 * ss4k_dev.h - measurement-only entry points of libss4k_hip_dev.so (built with -DSS4K_DEV from the
 * same sources as libss4k_hip.so; a superset of include/ss4k.h).  Not part of the product library:
 * the instrumented / alternative-tile-shape instantiations of the conv kernel live only here.
 * Used by tools/stamp4.py, tools/traffic_ablate.py, tools/conv_sweep.py.
 */
#ifndef SS4K_DEV_H
#define SS4K_DEV_H
#include "ss4k.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ss4k_model_desc.flags bits only libss4k_hip_dev.so accepts: kernels of rounds 2-4 that are on no product route any more. */
enum {
  SS4K_DEV_MODEL_CHAIN = 128,      /* RRDBNet fp16: the RRDB body of every job as ONE persistent launch with per-tile hand-offs (csrc/conv_chain.hip);
                                      bit-identical to SS4K_MODEL_NO_DENSE | SS4K_MODEL_NO_WIDE.  Its asynchronous failure mode: ss4k_model_check */
  SS4K_DEV_MODEL_CONV5_RS = 16384, /* RRDBNet fp16: conv5 of every RDB on the register-stationary kernel (csrc/conv_rs.hip) */
  SS4K_DEV_MODEL_FLAGS_ALL = 128 | 16384
};

/* Times ONE 3x3 conv layer (cin0 [+ cin1 concat] -> cout) in isolation on random operands: average
 * microseconds per launch over `iters` launches.
 * flags: 0 = the production kernel for that shape;
 *        32 (DBG_STAMP) = phase stamps (s_memtime per phase, printed to stderr), optionally combined with
 *        the timing-only ablations 1 (no output stores), 2 (every DMA reads one hot line), 16 (with 1:
 *        halo tiles from a 2 MB L2-resident window) - results are garbage in those builds;
 *        | shape_id << 8 selects another compiled tile shape (conv_mfma.hip, launch_conv3x3);
 *        | 2048 makes it conv5 of an RDB (no activation, out = conv * 0.2 + x);
 *        | 4096 runs the register-stationary kernel (conv_rs.hip) where the layer shape is built for it. */
int ss4k_bench_conv(ss4k_ctx* ctx, int dtype, int cin0, int cin1, int cout, int n, int h, int w, int flags,
                    int iters, double* avg_us, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SS4K_DEV_H */
