"""Test infrastructure (long-running, run by hand on a GPU box): random-configuration fuzz of the whole hot path (ss4k_upscale_frames, fp32 nets) against
the oracle service: uint8 frames within 1 LSB on <= 2 % of the bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
from oracle import nets as onets, service as osvc
from tests.helpers import assert_u8_close, smooth_u8

ctx = _capi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
bs_tab = W.bsvd_table(seed=21)
dn = factory.build_denoise_model(ctx, weights=bs_tab, dtype="f32")
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    single = bool(rng.integers(0, 2))
    n = int(rng.integers(1, 4))
    lh, lw = int(rng.integers(6, 30)) * 4, int(rng.integers(6, 40)) * 4  # BSVD needs multiples of 4
    h, w = (lh, lw) if rng.integers(0, 2) else (lh + int(rng.integers(0, 40)), lw + int(rng.integers(0, 60)))
    lrhr = bool(rng.integers(0, 4) > 0)
    up_f = int(rng.choice([2, 4]))
    if single:
        tab = W.fsrcnn_table(seed=3)
        sr = factory.build_model_fsrcnn(ctx, factor=up_f, weights=tab)
        model = lambda x: onets.fsrcnn(x, tab, up_f)
        mode = "fsrcnn"
    else:
        tab = W.srvgg_table(5, num_feat=16, num_conv=2, upscale=up_f)
        sr = _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F32, scale=up_f, num_feat=16, num_block=2), W.flatten(tab, W.srvgg_keys(2)))
        model = lambda x: onets.srvgg(x, tab, 2, up_f)
        mode = "realesrgan"
    denoise = single and bool(rng.integers(0, 2))
    out_shape = None
    if rng.integers(0, 2):
        out_shape = (int(rng.integers(lh, lh * up_f + 9)), int(rng.integers(lw, lw * up_f + 13)))
    if not single and not lrhr:
        h, w = lh, lw  # batched path without resize takes frames as they are; keep BSVD-free shapes general
    frames = torch.from_numpy(smooth_u8(100 + it, (n, h, w, 3)))
    rate = float(rng.choice([0.3, 1.0]))
    cfg = f"single={single} denoise={denoise} n={n} in={h}x{w} lr={lh}x{lw} x{up_f} out={out_shape} lr_hr_resize={lrhr}"
    try:
        up = _capi.Upscaler(ctx, sr, (lh, lw), out_shape, lrhr, single, dn if denoise else None, rate)
        osv = osvc.OracleUpscaler(model, denoising=denoise, denoise_rate=rate, upscaler_model=mode, lr_hr_resize=lrhr,
                                  denoise_model=lambda x: onets.bsvd_f1(x, bs_tab), output_shape=out_shape,
                                  single_mode=single, lr_shape=(lh, lw))
        for job in range(2):
            assert_u8_close(up(frames.cuda()), osv.upscale(frames), what=cfg + f" job{job}")
        print("ok ", cfg, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("BAD", cfg, repr(e)[:300], flush=True)
print("failures:", bad)
