"""GPU: the HIP path (through the C ABI) against the oracle and the reference-generated goldens.

fp32 path: rtol 1e-3 / atol 1e-4 (BASELINE.json north_star).  uint8 frames: at most one LSB
(truncation of floats that agree to 1e-4 can flip the integer).  fp16 path: PSNR.
"""
import os

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.upscale import model as factory
from oracle import nets as onets
from oracle import service as osvc
from tests.conftest import load_golden, manifest
from tests.helpers import assert_close, assert_u8_close, psnr, record_measured, rrdb_small_table, smooth_u8, srvgg_full_table, srvgg_table_for
from tests.test_oracle_golden import _t91, oracle_service_from_manifest

pytestmark = pytest.mark.gpu
CASES = manifest()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


# ------------------------------------------------------------------------------ glue ops (KAT)
def test_ops_known_answers(ctx):
    g = load_golden("kat_resample")
    x = dev(g["x"])
    assert_close(ctx.area_resize(x, (9, 14)), g["area_9x14"], what="area down")
    assert_close(ctx.area_resize(x, (23, 37)), g["area_23x37"], what="area identity")
    assert_close(ctx.area_resize(x, (30, 50)), g["area_30x50"], what="area up")
    assert_close(ctx.bicubic_resize(x, (31, 50)), g["bicubic_31x50"], what="bicubic up")
    assert_close(ctx.bicubic_resize(x, (11, 19)), g["bicubic_11x19"], what="bicubic down")
    assert_close(ctx.bilinear_resize(x, (46, 80)), g["bilinear_46x80"], what="bilinear")
    assert_close(ctx.depthwise_reflect(x, g["blur17_weight"]), g["blur17"], what="blur17 reflect")
    assert_close(ctx.depthwise_reflect(x, osvc.sharpen_kernel2d(0.00007).numpy()), g["sharpen_hr"], what="sharpen")


def test_ops_u8_and_stats(ctx):
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8)
    f = ctx.u8nhwc_to_f32nchw(dev(u8))
    want = torch.from_numpy(u8).permute(0, 3, 1, 2) / 255.0
    assert torch.equal(f.cpu(), want)  # exact: one fp32 division
    x = torch.rand(2, 3, 61, 47) * 0.7 + 0.1
    st = ctx.plane_stats(x.cuda()).cpu()
    assert_close(st[..., 0], x.reshape(2, 3, -1).mean(-1), rtol=1e-5, atol=1e-6, what="mean")
    assert_close(st[..., 1], x.reshape(2, 3, -1).std(-1), rtol=1e-5, atol=1e-6, what="unbiased std")
    y = torch.rand(1, 3, 20, 30) * 1.2 - 0.1  # exercises the clamp and the truncation
    got = ctx.f32nchw_to_u8nhwc(y.cuda()).cpu()
    want = (torch.clamp(y, 0, 1) * 255).permute(0, 2, 3, 1).to(torch.uint8)
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------ FSRCNN
@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("tag", ["t91", "gen"])
def test_fsrcnn_golden(ctx, factor, tag):
    g = load_golden(f"fsrcnn_x{factor}_{tag}")
    table = _t91(factor) if tag == "t91" else W.fsrcnn_table(seed=factor)
    m = factory.build_model_fsrcnn(ctx, factor=factor, weights=table)
    assert_close(m(dev(g["x"])), g["y"], what=f"fsrcnn x{factor} {tag}")


def test_config0_fsrcnn_x2_t91_256x256_service_vs_oracle(ctx):
    """BASELINE configs[0]'s workload at its stated size on the HIP path: FSRCNN x2, real T91 weights, ONE 256x256 random
    frame through ss4k_upscale_frames (per-frame path: the network runs on the three colour planes,
    fsrcnn_upscaler.py:292-299, model/fsrcnn/model.py:55-62) against the oracle service."""
    table = _t91(2)
    sr = factory.build_model_fsrcnn(ctx, factor=2, weights=table)
    up = _capi.Upscaler(ctx, sr, (256, 256), None, True, True, None, 1.0)
    frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1, 256, 256, 3), dtype=np.uint8))
    up.enable_taps(True)
    got = up(frames.cuda())
    osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=(256, 256))
    taps = {}
    want = osv.upscale_single(frames[0], taps)[None]
    assert got.shape == (1, 512, 512, 3)
    assert_u8_close(got, want, what="configs[0] FSRCNN x2 T91 256x256")
    assert_close(up.read_tap(1)[0], taps["model"][:, 0], what="configs[0] model tap")     # network output, fp32 tolerance
    assert_close(up.read_tap(4)[0], taps["final"][:, 0], what="configs[0] final float")


@pytest.mark.parametrize("factor", [2, 4])
def test_fsrcnn_split_tail_vs_exact(ctx, monkeypatch, factor):
    """The production tail computes its two products on the fp16 matrix rate from hi/lo-split operands (three MFMAs per
    product, fp32 accumulation) and overlap-adds in registers; SS4K_FS_EXACT=1 selects the exact-fp32 MFMA tail with the
    LDS ring.  On the real T91 checkpoint values the two agree to a few 1e-6 of the output peak - two orders inside the
    path's tolerance (rtol 1e-3 / atol 1e-4), which both also meet against the oracle."""
    table = _t91(factor)
    x = torch.rand(3, 1, 150, 333, generator=torch.Generator().manual_seed(4))   # ragged: several wave strips, partial last one
    monkeypatch.setenv("SS4K_FS_EXACT", "1")   # read when the model is built
    exact = factory.build_model_fsrcnn(ctx, factor=factor, weights=table)(x.cuda()).cpu()
    monkeypatch.delenv("SS4K_FS_EXACT")
    split = factory.build_model_fsrcnn(ctx, factor=factor, weights=table)(x.cuda()).cpu()
    peak = float(exact.abs().max())
    err = float((split - exact).abs().max())
    print(f"fsrcnn x{factor} T91: split-fp16 tail vs exact-fp32 tail max |diff| {err:.3g} = {err / peak:.2e} of the output peak {peak:.3g}")
    assert err <= 2e-5 * max(1.0, peak)
    with torch.no_grad():
        want = onets.fsrcnn(x, table, factor)
    assert_close(split, want, what=f"fsrcnn x{factor} split tail vs oracle")
    assert_close(exact, want, what=f"fsrcnn x{factor} exact tail vs oracle")


@pytest.mark.parametrize("shape", [(1, 1, 5, 7), (3, 1, 33, 65), (2, 1, 64, 31)])
def test_fsrcnn_ragged_shapes(ctx, shape):
    table = W.fsrcnn_table(seed=9)
    m = factory.build_model_fsrcnn(ctx, factor=2, weights=table)
    x = torch.rand(*shape)
    with torch.no_grad():
        want = onets.fsrcnn(x, table, 2)
    assert_close(m(x.cuda()), want, what=f"fsrcnn {shape}")


# ------------------------------------------------------------------------------ SRVGG
@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("srvgg_")])
def test_srvgg_golden_fp32(ctx, name):
    mm = CASES[name]
    g = load_golden(name)
    table = srvgg_table_for(mm)
    desc = _capi.make_desc(_capi.SRVGG, _capi.F32, scale=mm["upscale"], num_feat=mm["num_feat"], num_block=mm["num_conv"])
    m = _capi.Model(ctx, desc, W.flatten(table, W.srvgg_keys(mm["num_conv"])))
    assert_close(m(dev(g["x"])), g["y"], what=name)


def test_srvgg_fp16_psnr(ctx):
    mm = CASES["srvgg_f64_c4_x4"]
    g = load_golden("srvgg_f64_c4_x4")
    table = W.srvgg_table(seed=11, num_feat=64, num_conv=4, upscale=4)
    desc = _capi.make_desc(_capi.SRVGG, _capi.F16, scale=4, num_feat=64, num_block=4)
    m = _capi.Model(ctx, desc, W.flatten(table, W.srvgg_keys(4)))
    p = psnr(m(dev(g["x"])), g["y"])
    record_measured("srvgg_f64_c4_x4_fp16_vs_reference_golden", psnr_db=p, asserted="PSNR > 85.5 dB (peak 1.0)")
    assert p > 85.5, p   # measured 87.5 dB (profiles/earlier/r05/r05_parity_measured.json), asserted at - 2 dB; the 32-conv default: tests/test_gpu_srvgg_default.py


# ------------------------------------------------------------------------------ BSVD
@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("bsvd32_")])
def test_bsvd_golden_fp32(ctx, name):
    g = load_golden(name)
    # bsvd32_f1_*: one independent frame per call (the service); bsvd32_seq*: (N,F,4,H,W) clips run as
    # one stream through the bidirectional buffers (BSVD.forward)
    m = factory.build_denoise_model(ctx, weights=W.bsvd_table(seed=21), dtype="f32", stream="_seq" in name)
    y = m(dev(g["x"]))
    assert y.shape == g["y"].shape
    assert_close(y, g["y"], what=name)


@pytest.mark.parametrize("name", ["bsvd32_f1_32x48", "bsvd32_seq5_24x40"])
def test_bsvd_fp16_psnr(ctx, name):
    g = load_golden(name)
    m = factory.build_denoise_model(ctx, weights=W.bsvd_table(seed=21), dtype="f16", stream="_seq" in name)
    assert psnr(m(dev(g["x"])), g["y"]) > 45.0


@pytest.mark.parametrize("stream", [False, True])
def test_bsvd64_variant_vs_oracle(ctx, stream):
    """The wider BSVD the reference's factory also knows (chns 64/128/256, mid 64, interm 64:
    bsvd/factory.py:26-30,94-98), per frame and as a stream, fp32 against the oracle."""
    kw = factory.BSVD_VARIANTS["bsvd-64"]
    tab = W.bsvd_table(seed=33, **kw)
    x = torch.rand(1, 3, 4, 24, 40) if stream else torch.rand(2, 1, 4, 24, 40)
    x[:, :, 3] = 0.05
    with torch.no_grad():
        want = (onets.bsvd_seq if stream else onets.bsvd_f1)(x, tab)
    m = factory.build_denoise_model(ctx, weights=tab, dtype="f32", stream=stream, variant="bsvd-64")
    assert_close(m(x.cuda()), want, what=f"bsvd-64 stream={stream}")
    m16 = factory.build_denoise_model(ctx, weights=tab, dtype="f16", stream=stream, variant="bsvd-64")
    assert psnr(m16(x.cuda()), want) > 40.0


def test_bsvd_stream_properties(ctx):
    """Size-independent checks of the stream mode at a larger frame: a one-frame stream equals the
    per-frame model; every frame of a stream depends on its neighbours (the temporal shift is live);
    splitting a stream changes only the frames next to the cut... at depth 16 that is all of a short
    clip, so check the weaker, exact property: frame order matters and results are deterministic."""
    tab = W.bsvd_table(seed=21)
    ms = factory.build_denoise_model(ctx, weights=tab, dtype="f32", stream=True)
    m1 = factory.build_denoise_model(ctx, weights=tab, dtype="f32", stream=False)
    x = torch.rand(1, 4, 4, 96, 160)
    x[:, :, 3] = 0.05
    ys = ms(x.cuda())
    assert ys.shape == (1, 4, 3, 96, 160)
    assert torch.equal(ys, ms(x.cuda()))
    one = ms(x[:, :1].cuda())
    assert torch.equal(one[:, 0], m1(x[:, 0].cuda()))          # F = 1 degenerates to the masked conv net
    assert (ys[:, 0] - one[:, 0]).abs().max() > 1e-4            # frame 0 sees frame 1 through the buffers
    assert (ms(x.flip(1).cuda()).flip(1) - ys).abs().max() > 1e-4  # left and right folds are different channels


# ------------------------------------------------------------------------------ RRDBNet (oracle unpinned, see oracle/__init__.py)
@pytest.mark.parametrize("scale,shape", [(2, (1, 3, 32, 48)), (2, (2, 3, 20, 72)), (4, (1, 3, 16, 40)), (1, (1, 3, 32, 64))])
def test_rrdbnet_fp32_vs_oracle(ctx, scale, shape):
    table = rrdb_small_table(seed=5 + scale, scale=scale, num_block=2)
    m = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=table, dtype="f32", scale=scale, num_block=2)
    x = torch.rand(*shape)
    with torch.no_grad():
        want = onets.rrdbnet(x, table, scale, 2)
    assert_close(m(x.cuda()), want, what=f"rrdbnet x{scale} {shape}")


def test_rrdbnet_fp16_psnr(ctx):
    table = rrdb_small_table(seed=7, scale=2, num_block=3)
    m = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=table, dtype="f16", num_block=3)
    x = torch.rand(1, 3, 48, 64)
    with torch.no_grad():
        want = onets.rrdbnet(x, table, 2, 3)
    assert psnr(m(x.cuda()), want) > 50.0


# ------------------------------------------------------------------------------ service glue
def _hip_service_from_manifest(ctx, m, dtype="f32"):
    if m["sr"] == "srvgg":
        t = srvgg_table_for(m)
        desc = _capi.make_desc(_capi.SRVGG, _capi.F32 if dtype == "f32" else _capi.F16, scale=m["upscale"],
                               num_feat=m["num_feat"], num_block=m["num_conv"])
        sr = _capi.Model(ctx, desc, W.flatten(t, W.srvgg_keys(m["num_conv"])))
    else:
        sr = factory.build_model_fsrcnn(ctx, factor=m["factor"], weights=W.fsrcnn_table(seed=m["seed"]))
    single = m["single_mode"] if m["single_mode"] is not None else (m["mode"] != "realesrgan")
    dn = factory.build_denoise_model(ctx, weights=W.bsvd_table(seed=m["bsvd_seed"]), dtype=dtype) if m["denoising"] else None
    up = _capi.Upscaler(ctx, sr, m["lr_shape"], m["output_shape"], m["lr_hr_resize"], single, dn, m["denoise_rate"])
    return up, (sr, dn)


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("svc_")])
def test_service_golden_u8(ctx, name):
    """uint8 frames out of the HIP path vs the frames the REFERENCE service produced."""
    g = load_golden(name)
    up, keep = _hip_service_from_manifest(ctx, CASES[name])
    frames = dev(g["frames"])
    assert_u8_close(up(frames), g["out1"], what=name + " job 1")
    assert_u8_close(up(frames), g["out2"], what=name + " job 2")


@pytest.mark.parametrize("name", ["svc_multi_srvgg_x4_color", "svc_multi_srvgg_x4_area_bicubic", "svc_multi_srvgg64x32_x4_area_bicubic",
                                  "svc_single_fsrcnn_x2_denoise", "svc_single_srvgg_x2_denoise"])
def test_service_float_taps_vs_oracle(ctx, name):
    m = CASES[name]
    g = load_golden(name)
    up, keep = _hip_service_from_manifest(ctx, m)
    up.enable_taps(True)
    up(dev(g["frames"]))
    osv = oracle_service_from_manifest(m)
    frames = torch.from_numpy(g["frames"])
    single = m["single_mode"] if m["single_mode"] is not None else (m["mode"] != "realesrgan")
    if single:
        taps = [dict() for _ in range(frames.shape[0])]
        for i in range(frames.shape[0]):
            osv.upscale_single(frames[i], taps[i])
        want = {k: torch.stack([t[k] for t in taps]) for k in ("lr", "model", "final")}
        want["model"] = want["model"][:, :, 0]
        want["final"] = want["final"][:, :, 0]
        pairs = [(0, "lr"), (1, "model"), (4, "final")]
    else:
        t = {}
        osv.upscale_multi(frames, t)
        want = t
        pairs = [(0, "lr"), (1, "model"), (2, "stats"), (3, "color"), (4, "final")]
    for which, key in pairs:
        assert_close(up.read_tap(which), want[key], what=f"{name} tap {key}")


def test_service_rejects_bad_input(ctx):
    sr = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2))
    up = _capi.Upscaler(ctx, sr, (16, 16), None, True, True, None, 1.0)
    with pytest.raises(AssertionError):
        up(torch.zeros(1, 16, 16, 4, dtype=torch.uint8, device="cuda"))
    with pytest.raises(_capi.Ss4kError):
        _capi.Upscaler(ctx, sr, (16, 16), None, True, False, None, 1.0)  # batched path needs a 3-channel SR model
    with pytest.raises(_capi.Ss4kError):
        _capi.Model(ctx, _capi.make_desc(_capi.FSRCNN, _capi.F32, scale=2), np.zeros(10, np.float32))
    with pytest.raises(_capi.Ss4kError):  # stream mode is a BSVD property
        d = _capi.make_desc(_capi.SRVGG, _capi.F32, scale=2, num_feat=16, num_block=2, bsvd_stream=True)
        _capi.Model(ctx, d, W.flatten(W.srvgg_table(0, num_feat=16, num_conv=2, upscale=2), W.srvgg_keys(2)))
    with pytest.raises(_capi.Ss4kError):  # RRDBNet x2 needs even sizes (pixel_unshuffle raises in the reference too)
        factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=rrdb_small_table(seed=7, scale=2, num_block=1),
                                   dtype="f32", scale=2, num_block=1)(torch.rand(1, 3, 15, 16).cuda())


# ------------------------------------------------------------------------------ ragged shapes
@pytest.mark.parametrize("seed", [1, 2])
def test_conv_networks_random_shapes(ctx, seed):
    """Edge tiles of the conv kernel: random N, H, W (not multiples of the 16x32 tile, smaller than a
    tile, several frames) through one-block RRDBNet x1/x2/x4, a small SRVGG and BSVD; fp32 path within
    the parity tolerance of the CPU oracle, fp16 path by PSNR.  (tests/fuzz_shapes.py runs more.)"""
    rng = np.random.default_rng(seed)
    bs_tab = W.bsvd_table(seed=21)
    sv_tab = W.srvgg_table(5, num_feat=32, num_conv=3, upscale=2)
    for it in range(8):
        kind = ("rrdb", "srvgg", "bsvd", "rrdb")[it % 4]
        n = int(rng.integers(1, 4))
        if kind == "rrdb":
            sc = int(rng.choice([1, 2, 4])); r = {1: 4, 2: 2, 4: 1}[sc]
            x = torch.rand(n, 3, int(rng.integers(1, 24)) * r, int(rng.integers(1, 44)) * r)
            tab = W.rrdbnet_table(11 + sc, scale=sc, num_feat=64, num_block=1, num_grow_ch=32)
            with torch.no_grad():
                want = onets.rrdbnet(x, tab, sc, 1)
            build = lambda d: factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=tab, dtype=d, scale=sc, num_block=1)
        elif kind == "srvgg":
            x = torch.rand(n, 3, int(rng.integers(1, 70)), int(rng.integers(1, 110)))
            with torch.no_grad():
                want = onets.srvgg(x, sv_tab, 3, 2)
            build = lambda d: _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F32 if d == "f32" else _capi.F16, scale=2,
                                                               num_feat=32, num_block=3), W.flatten(sv_tab, W.srvgg_keys(3)))
        else:
            x = torch.rand(n, 1, 4, int(rng.integers(1, 14)) * 4, int(rng.integers(1, 20)) * 4)
            x[:, :, 3] = 0.05
            with torch.no_grad():
                want = onets.bsvd_f1(x, bs_tab)
            build = lambda d: factory.build_denoise_model(ctx, weights=bs_tab, dtype=d)
        what = f"{kind} {tuple(x.shape)}"
        assert_close(build("f32")(x.cuda()), want, what=what)
        assert psnr(build("f16")(x.cuda()), want) > 40.0, what


def test_service_random_configurations(ctx):
    """ss4k_upscale_frames on random configurations (batched / per-frame, +-denoise, +-area pre-resize,
    +-bicubic output, x2 / x4, odd frame sizes, first and later job) against the oracle service.
    tests/fuzz_service.py runs the long version."""
    rng = np.random.default_rng(17)
    bs_tab = W.bsvd_table(seed=21)
    dn = factory.build_denoise_model(ctx, weights=bs_tab, dtype="f32")
    for it in range(10):
        single = bool(it % 2)
        n = int(rng.integers(1, 4))
        lh, lw = int(rng.integers(6, 24)) * 4, int(rng.integers(6, 32)) * 4
        h, w = (lh, lw) if rng.integers(0, 2) else (lh + int(rng.integers(0, 40)), lw + int(rng.integers(0, 60)))
        lrhr = bool(rng.integers(0, 4) > 0)
        f = int(rng.choice([2, 4]))
        if single:
            tab = W.fsrcnn_table(seed=3)
            sr = factory.build_model_fsrcnn(ctx, factor=f, weights=tab)
            net, mode = (lambda x, tab=tab, f=f: onets.fsrcnn(x, tab, f)), "fsrcnn"
        else:
            tab = W.srvgg_table(5, num_feat=16, num_conv=2, upscale=f)
            sr = _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F32, scale=f, num_feat=16, num_block=2),
                             W.flatten(tab, W.srvgg_keys(2)))
            net, mode = (lambda x, tab=tab, f=f: onets.srvgg(x, tab, 2, f)), "realesrgan"
            if not lrhr:
                h, w = lh, lw
        denoise = single and bool(rng.integers(0, 2))
        out_shape = (int(rng.integers(lh, lh * f + 9)), int(rng.integers(lw, lw * f + 13))) if rng.integers(0, 2) else None
        rate = float(rng.choice([0.3, 1.0]))
        frames = torch.from_numpy(smooth_u8(200 + it, (n, h, w, 3)))
        what = f"single={single} denoise={denoise} n={n} in={h}x{w} lr={lh}x{lw} x{f} out={out_shape} resize={lrhr}"
        up = _capi.Upscaler(ctx, sr, (lh, lw), out_shape, lrhr, single, dn if denoise else None, rate)
        osv = osvc.OracleUpscaler(net, denoising=denoise, denoise_rate=rate, upscaler_model=mode, lr_hr_resize=lrhr,
                                  denoise_model=lambda x: onets.bsvd_f1(x, bs_tab), output_shape=out_shape,
                                  single_mode=single, lr_shape=(lh, lw))
        for job in range(2):
            assert_u8_close(up(frames.cuda()), osv.upscale(frames), what=f"{what} job {job}")


# ------------------------------------------------------------------------------ image-server mode
_IMG_MODE = {"sr": "srvgg", "seed": 41, "num_feat": 32, "num_conv": 2, "upscale": 4, "mode": "realesrgan",
             "lr_shape": [360, 640], "output_shape": None, "lr_hr_resize": False, "denoising": False,
             "denoise_rate": 0.2, "single_mode": None, "bsvd_seed": 21}


@pytest.mark.parametrize("hw", [(61, 83), (129, 257), (203, 99)])
def test_image_mode_any_shape_vs_oracle(ctx, hw):
    """The HTTP image path's service configuration (image_pipeline.py:58-63: realesrgan, batch_size=1,
    lr_hr_resize=False, denoising off) on one (1,H,W,3) frame of arbitrary, odd size
    (image_pipeline.py:280-287): no resize to lr_shape, x4 network, stats + colour match, truncation."""
    up, keep = _hip_service_from_manifest(ctx, _IMG_MODE)
    frame = torch.from_numpy(smooth_u8(50 + hw[0], (1, hw[0], hw[1], 3)))
    got = up(frame.cuda())
    assert got.shape == (1, 4 * hw[0], 4 * hw[1], 3)
    want = oracle_service_from_manifest(_IMG_MODE).upscale(frame)
    assert_u8_close(got, want.numpy(), what=f"image mode {hw}")


def test_image_mode_max_size_runs(ctx):
    """Largest frame the image server admits (4096x2048 pixels, image_pipeline.py:265-271) through the
    shipped default net (SRVGG x4, fp16): 16384x8192 out.  Size-independent checks: the channel
    statistics match makes every output channel's mean equal the input's (within truncation), and
    the call is deterministic."""
    sr = factory.build_model_esrgan(ctx, "realesr-general-x4v3", weights="synthetic", dtype="f16", seed=3)
    up = _capi.Upscaler(ctx, sr, (360, 640), None, False, False, None, 0.2)
    frame = torch.from_numpy(smooth_u8(77, (1, 2048, 4096, 3))).cuda()
    out = up(frame)
    assert out.shape == (1, 8192, 16384, 3) and out.dtype == torch.uint8
    m_in = frame.float().mean(dim=(0, 1, 2))
    m_out = out[:, ::4, ::4].float().mean(dim=(0, 1, 2))
    assert (m_in - m_out).abs().max() < 1.5, (m_in, m_out)
    assert torch.equal(out[:, :64], up(frame)[:, :64])
    del out
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------ full size (BASELINE configs), size-independent properties
@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_config1_fsrcnn_x2_t91_720p_service_vs_oracle(ctx, dtype):
    """BASELINE configs[1] at its full size with the REAL checkpoint: FSRCNN x2 (T91), one 720p frame -> 1440p through
    ``ss4k_upscale_frames`` (per-frame path on the three colour planes, statistics match, truncation) against the oracle service, in
    both arithmetic modes: fp32-grade (uint8 within 1 LSB, float taps at the literal tolerance) and the reference engine's fp16
    (uint8 within 1 LSB, PSNR recorded)."""
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    table = _t91(2)
    sr = factory.build_model_fsrcnn(ctx, factor=2, weights=table, dtype=dtype)
    up = _capi.Upscaler(ctx, sr, (720, 1280), None, True, True, None, 1.0)
    frames = torch.from_numpy(smooth_u8(41, (1, 720, 1280, 3)))
    if dtype == "f32":
        up.enable_taps(True)
    got = up(frames.cuda()).cpu()
    osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=(720, 1280))
    taps = {}
    want = osv.upscale_single(frames[0], taps)[None]
    assert got.shape == (1, 1440, 2560, 3) and got.dtype == torch.uint8
    d = (got.int() - want.int()).abs()
    p = psnr(got.float(), want.float(), peak=255.0)
    frac = float((d > 0).float().mean())
    record_measured(f"config1_fsrcnn_x2_t91_720p_{dtype}_service", psnr_db=p, max_lsb=int(d.max()), bytes_differ=frac)
    print(f"configs[1] FSRCNN x2 T91 720p {dtype}: PSNR {p:.2f} dB, max {int(d.max())} LSB, {frac:.4%} of the bytes differ")
    if dtype == "f32":
        assert_u8_close(got, want, what="configs[1] FSRCNN x2 T91 720p")
        assert_close(up.read_tap(1)[0], taps["model"][:, 0], what="configs[1] model tap")
        assert_close(up.read_tap(4)[0], taps["final"][:, 0], what="configs[1] final float")
    else:
        assert int(d.max()) <= 1 and frac < 0.08 and p > 55.0, (int(d.max()), frac, p)


def test_fsrcnn_720p_properties(ctx):
    """C2 size: linearity in the deconv bias and per-plane independence, plus a sampled-window oracle check."""
    table = W.fsrcnn_table(seed=2)
    m = factory.build_model_fsrcnn(ctx, factor=2, weights=table)
    x = torch.rand(3, 1, 720, 1280)
    y = m(x.cuda()).cpu()
    assert y.shape == (3, 1, 1440, 2560) and torch.isfinite(y).all()
    # plane independence: permuting input planes permutes outputs
    y2 = m(x[[2, 0, 1]].cuda()).cpu()
    assert torch.equal(y2, y[[2, 0, 1]])
    # window check against the oracle: receptive field radius is 2+4+2 = 8 LR px < 24 px margin
    win = x[:, :, 300:396, 500:628]
    with torch.no_grad():
        want = onets.fsrcnn(win, table, 2)
    assert_close(y[:, :, 600 + 48:792 - 48, 1000 + 48:1256 - 48], want[:, :, 48:-48, 48:-48], what="720p window")


def test_rrdbnet_720p_fp16_vs_fp32_psnr(ctx):
    """C3 size, 23 blocks: fp16 path vs the fp32 path of the same library (PSNR), output statistics sane."""
    table = W.rrdbnet_table(0, scale=2)
    flat = W.flatten(table, W.rrdbnet_keys(23))
    m16 = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), flat)
    m32 = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F32, scale=2), flat)
    x = torch.from_numpy(smooth_u8(1, (1, 720, 1280, 3))).permute(0, 3, 1, 2).float().div(255.0).cuda()
    y16 = m16(x)
    y32 = m32(x)
    assert y16.shape == (1, 3, 1440, 2560)
    assert torch.isfinite(y16).all() and torch.isfinite(y32).all()
    p = psnr(y16, y32, peak=float(y32.abs().max()))
    assert p > 45.0, f"fp16 vs fp32 PSNR {p:.1f} dB"
    # window check of the fp32 path against the oracle with a reduced receptive field is not
    # possible (RRDB receptive field ~ 350 px), so compare a 2-block model on a crop instead
    t2 = rrdb_small_table(seed=11, scale=2, num_block=2)
    ms = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=t2, dtype="f32", num_block=2)
    xs = x[:, :, :96, :160].cpu()
    with torch.no_grad():
        want = onets.rrdbnet(xs, t2, 2, 2)
    assert_close(ms(xs.cuda()), want, what="rrdbnet crop")


# ------------------------------------------------------------------------------ drop-in service, worker process on the GPU
def test_hip_service_worker_process_roundtrip():
    """HipUpscalerService behind the reference's queue API: start() spawns the worker, frames go in as
    device tensors inside UpscalerQueueEntry, upscaled uint8 frames come back; checked against the oracle."""
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
    from sharkshark4k_amd.util import Profiler
    table = W.fsrcnn_table(seed=2)
    svc = HipUpscalerService(device=0, denoising=False, upscaler_model="fsrcnn", scale=2, lr_shape=(36, 52),
                             weights={"sr": table})
    svc.start()
    try:
        frames = torch.from_numpy(smooth_u8(45, (2, 36, 52, 3)))
        for step in (0, 1):
            svc.push_job(UpscalerQueueEntry(frames=frames, step=step, profiler=Profiler()), timeout=60)
        res = [svc.get_result(timeout=180) for _ in range(2)]
        assert [r.step for r in res] == [0, 1]
        got = res[0].frames.cpu()
        assert got.shape == (2, 72, 104, 3) and got.dtype == torch.uint8
        osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=(36, 52))
        assert_u8_close(got, osv.upscale(frames), what="service roundtrip")
        assert "upscaler.upscale" in res[0].profiler.data and res[0].elapsed > 0
    finally:
        svc.stop()
    assert not svc.proc.is_alive()


def test_rrdbnet_x4_plane_beyond_4gib(ctx):
    """RRDBNet x4 on the image server's largest frame (4096x2048, image_pipeline.py:265-271): the tail
    runs on 16384x8192 = 134 M pixels, 4.29 GB per fp16 plane - past 32-bit byte offsets.  Checked by
    shift invariance: far-corner crop of the big output == the net run on the matching input crop
    (interior only, one RRDB block has a receptive field of ~25 LR pixels)."""
    tab = W.rrdbnet_table(21, scale=4, num_feat=64, num_block=1, num_grow_ch=32)
    m = factory.build_model_esrgan(ctx, "RealESRGAN_x4plus", weights=tab, dtype="f16", scale=4, num_block=1)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 2048, 4096, generator=g)
    big = m(x.cuda())
    assert big.shape == (1, 3, 8192, 16384)
    y0, x0, sz, mg = 1760, 3808, 256, 40  # crop near the highest addresses; mg = margin in LR pixels
    small = m(x[:, :, y0:y0 + sz, x0:x0 + sz].contiguous().cuda())
    a = big[:, :, 4 * (y0 + mg):4 * (y0 + sz - mg), 4 * (x0 + mg):4 * (x0 + sz - mg)]
    b = small[:, :, 4 * mg:4 * (sz - mg), 4 * mg:4 * (sz - mg)]
    assert torch.isfinite(big[:, :, ::64, ::64]).all()
    assert (a - b).abs().max().item() < 2e-3, (a - b).abs().max().item()
    del big, small
    torch.cuda.empty_cache()


def test_service_profiler_keys_match_reference():
    """The worker's result carries the reference's span keys for the same configuration
    (MANIFEST profiler_keys recorded from the reference service): 'fsrcnn.denoise' only when the job
    denoised, 'fsrcnn.model', 'upscaler.upscale'."""
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
    from sharkshark4k_amd.util import Profiler
    for name in ("svc_single_fsrcnn_x2_denoise", "svc_single_fsrcnn_x2"):
        m = CASES[name]
        svc = HipUpscalerService(device=0, denoising=m["denoising"], denoise_rate=m["denoise_rate"], upscaler_model="fsrcnn",
                                 scale=m["factor"], lr_shape=tuple(m["lr_shape"]), dtype="f32",
                                 weights={"sr": W.fsrcnn_table(seed=m["seed"]), "denoise": W.bsvd_table(seed=m["bsvd_seed"])})
        svc.proc_init()  # in-process: the worker body without the process around it
        g = load_golden(name)
        res = svc.proc_job_recieved(UpscalerQueueEntry(frames=dev(g["frames"]), step=0, profiler=Profiler()))
        assert_u8_close(res.frames, g["out1"], what=name)
        assert sorted(k for k in res.profiler.data) == sorted(m["profiler_keys"]), (name, res.profiler.data)


def test_model_workspace_query(ctx):
    """ss4k_model_workspace_bytes: the activation bytes a forward of that shape will hold, computed
    without touching the device; checked against what the forward then really allocates."""
    m = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=rrdb_small_table(seed=9, scale=2, num_block=1),
                                   dtype="f16", scale=2, num_block=1)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    want = m.workspace_bytes(2, 360, 640)
    assert torch.cuda.mem_get_info()[0] == free0  # the query allocates nothing
    assert want > 0 and m.workspace_bytes(4, 360, 640) > want > m.workspace_bytes(1, 360, 640)
    x = torch.rand(2, 3, 360, 640, device="cuda")
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    y = m(x)
    torch.cuda.synchronize()
    used = free1 - torch.cuda.mem_get_info()[0] - y.numel() * 4
    assert abs(used - want) <= 64 * 2 ** 20, (used, want)  # allocator granularity (2 MiB pages per buffer)


def test_stream_dispatcher_over_two_hip_services():
    """SURVEY §8 f1 on the device: a recorder batch cut into 4-frame jobs, dealt round-robin over two
    HipUpscalerService worker processes (both on GPU 0 here; one per GPU on a node), results re-ordered
    by step and checked against the oracle service."""
    from sharkshark4k_amd.stream import StreamDispatcher
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    table = W.fsrcnn_table(seed=2)
    svcs = [HipUpscalerService(device=0, denoising=False, upscaler_model="fsrcnn", scale=2, lr_shape=(36, 52),
                               weights={"sr": table}) for _ in range(2)]
    for s_ in svcs:
        s_.start()
    try:
        d = StreamDispatcher(svcs, fps=24, frame_skips=False)
        frames = torch.from_numpy(smooth_u8(91, (12, 36, 52, 3)))
        steps = d.submit_batch(frames)
        assert steps == [0, 1, 2]
        out = d.drain(steps, timeout=240)
        assert [e.step for e in out] == [0, 1, 2]
        osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=(36, 52))
        for e in out:
            assert_u8_close(e.frames.cpu(), osv.upscale(frames[4 * e.step:4 * e.step + 4]), what=f"stream job {e.step}")
            assert "upscaler.upscale.per_frame_ms" in e.profiler.data
    finally:
        for s_ in svcs:
            s_.stop()


# ------------------------------------------------------------------------------ fused service tails
@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("svc_")])
def test_fused_tails_are_bit_identical_to_the_unfused_path(ctx, name):
    """Without parity taps the service runs fused tails (statistics riding along with the producer, normalisation
    applied on the fly by its consumers, clamp / bicubic / uint8 in one pass).  Every per-element expression is
    shared with the one-kernel-per-torch-call path the taps use, so the frames must be IDENTICAL - and both are
    checked against the frames the reference produced."""
    g = load_golden(name)
    frames = dev(g["frames"])
    up_f, keep_f = _hip_service_from_manifest(ctx, CASES[name])
    up_u, keep_u = _hip_service_from_manifest(ctx, CASES[name])
    up_u.enable_taps(True)
    for job, key in enumerate(("out1", "out2")):
        a, b = up_f(frames), up_u(frames)
        assert torch.equal(a, b), f"{name} job {job}: fused and unfused frames differ in {int((a != b).sum())} bytes"
        assert_u8_close(a, g[key], what=f"{name} fused job {job}")


def test_fused_tails_full_size_identical(ctx):
    """Same at the shipped default's size (SRVGG x4 on 720p, bicubic to 1440p, colour match on) in fp16, where the
    statistics come out of the PixelShuffle tail's accumulators: mean / std may differ in the last fp64 bits of
    the sum, so frames are compared within 1 LSB and must be almost everywhere identical."""
    # (SS4K_MODEL_HR_F32: the fused path of an fp16 SRVGG would otherwise keep the HR tensor in fp16 - that difference has its own test,
    # test_srvgg_f16_half_hr_tensor_vs_fp32_hr_tensor; this one is about the fusion)
    sr = factory.build_model_esrgan(ctx, "realesr-general-x4v3", weights="synthetic", dtype="f16", seed=3, flags=_capi.MODEL_HR_F32)
    up_f = _capi.Upscaler(ctx, sr, (720, 1280), (1440, 2560), True, False, None, 0.5)
    up_u = _capi.Upscaler(ctx, sr, (720, 1280), (1440, 2560), True, False, None, 0.5)
    up_u.enable_taps(True)
    frames = torch.from_numpy(smooth_u8(12, (2, 720, 1280, 3))).cuda()
    a, b = up_f(frames), up_u(frames)
    d = (a.int() - b.int()).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-4, (int(d.max()), float((d > 0).float().mean()))


# ------------------------------------------------------------------------------ FSRCNN in fp16 (the reference engine's precision)
def _wild_slopes(table, seed):
    """Every PReLU slope redrawn from [-0.6, 1.8]: slopes above 1 (the real checkpoints have them), negative ones and ordinary ones.  The
    matrix-core modes compute PReLU as y + c |y| on accumulators of weights scaled by (1 + s) / 2 (models.cpp): c runs from - 0.29 to 4."""
    t = dict(table)
    rng = np.random.default_rng(seed)
    for k in list(t):
        if np.asarray(t[k]).ndim == 1 and k.endswith(".weight"):
            t[k] = rng.uniform(-0.6, 1.8, np.asarray(t[k]).shape).astype(np.float32)
    return t


@pytest.mark.parametrize("factor,shape", [(2, (3, 1, 70, 141)), (4, (2, 1, 45, 66))])
def test_fsrcnn_f32_grade_wild_slopes_vs_oracle(ctx, factor, shape):
    """fp32-grade mode with every PReLU slope redrawn from [-0.6, 1.8]: the matrix-core kernels take the one-fma form on scaled weights
    (models.cpp) - at the literal fp32 tolerance against the oracle, and against the exact-fp32 kernels, which run on the unscaled weights
    with the select form."""
    table = _wild_slopes(W.fsrcnn_table(seed=factor), 10 + factor)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[3]))
    with torch.no_grad():
        want = onets.fsrcnn(x, table, factor)
    got = factory.build_model_fsrcnn(ctx, factor=factor, weights=table)(x.cuda()).cpu()
    assert_close(got, want, what=f"fsrcnn x{factor} wild slopes, fp32-grade")
    exact = _capi.Model(ctx, _capi.make_desc(_capi.FSRCNN, _capi.F32, scale=factor, flags=_capi.MODEL_FS_EXACT), W.flatten(table, W.fsrcnn_keys()))(x.cuda()).cpu()
    assert_close(exact, want, what=f"fsrcnn x{factor} wild slopes, exact kernels")


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_fsrcnn_slope_near_minus_one_takes_the_exact_kernels(ctx, dtype):
    """The one-fma PReLU needs (1 + s) / 2 > 0 with room to spare: a checkpoint with a slope at or below -0.875 (here -0.95 and -1.3 on some
    channels; no real checkpoint has one) is run by the exact-fp32 kernels on unscaled weights, whatever dtype was asked for - fp32 tolerance."""
    table = dict(W.fsrcnn_table(seed=7))
    n = 0
    for k in list(table):
        v = np.asarray(table[k])
        if v.ndim == 1 and k.endswith(".weight"):
            v = v.copy(); v[0] = -0.95; v[-1] = -1.3
            table[k] = v; n += 1
    assert n == 7
    x = torch.rand(3, 1, 40, 77, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = onets.fsrcnn(x, table, 2)
    got = factory.build_model_fsrcnn(ctx, factor=2, weights=table, dtype=dtype)(x.cuda()).float().cpu()
    assert_close(got, want, what=f"fsrcnn x2, slopes near -1, dtype {dtype} (exact kernels)")


@pytest.mark.parametrize("factor,tag,shape", [(2, "t91", (3, 1, 150, 333)), (4, "t91", (3, 1, 97, 130)), (2, "syn", (12, 1, 64, 260)),
                                              (2, "syn", (1, 1, 5, 7)), (4, "syn", (2, 1, 33, 129)), (2, "t91", (1, 1, 256, 256)),
                                              (2, "wild", (3, 1, 70, 141)), (4, "wild", (2, 1, 45, 66))])
def test_fsrcnn_f16_mode_vs_oracle(ctx, factor, tag, shape):
    """dtype f16 (fp16 operands, fp32 accumulation, fp16 intermediates; head on MFMA with the bias in a spare K slot): judged by
    PSNR against the fp32 CPU forward, like the fp16 RRDBNet path; ragged shapes cover partial strips, bands and both tile parities."""
    table = _t91(factor) if tag == "t91" else W.fsrcnn_table(seed=factor) if tag == "syn" else _wild_slopes(W.fsrcnn_table(seed=factor), factor)
    if tag == "t91":   # the real checkpoints DO have slopes above 1 (x2: one channel at 1.04; x4: up to 9.1)
        assert max(float(np.max(v)) for k, v in table.items() if np.asarray(v).ndim == 1 and k.endswith(".weight")) > 1.0
    m = factory.build_model_fsrcnn(ctx, factor=factor, weights=table, dtype="f16")
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2]))
    with torch.no_grad():
        want = onets.fsrcnn(x, table, factor)
    got = m(x.cuda()).float().cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    peak = float(want.abs().max())
    p = psnr(got / peak, want / peak)
    err = float((got - want).abs().max())
    record_measured(f"fsrcnn_f16_x{factor}_{tag}_{shape[2]}x{shape[3]}", psnr_db=p, max_abs_err=err, peak=peak)
    print(f"fsrcnn f16 x{factor} {tag} {shape}: PSNR {p:.1f} dB, max |d| {err:.3g} of peak {peak:.3g}")
    # measured 69.1-85.8 dB over the six cases (profiles/earlier/r05/r05_parity_measured.json): asserted at the worst case - 2 dB
    assert p > 67.0 and err < 1e-2 * max(1.0, peak)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_fsrcnn_service_accumulators_clean_themselves_across_jobs(ctx, dtype):
    """The uint8-direct FSRCNN job (no area resize, no denoiser) keeps ONE set of fp64 accumulators for both statistics, and its finishing
    launch zeroes what it has read instead of the next job paying a memset (api.cpp: st_acc2).  Jobs of 4, 1, 3, 2, 4 frames through one
    upscaler - different plane counts, hence different accumulator layouts over the same buffer - against a NEW upscaler per job: every
    byte.  Then the same job CAPTURED into a graph on a side stream (a captured job always carries the memset: it runs later, in whatever
    state eager jobs in between leave) and replayed around eager jobs."""
    sr = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=5), dtype=dtype)
    lr_shape = (46, 84)
    frames = torch.from_numpy(smooth_u8(91, (4, lr_shape[0], lr_shape[1], 3))).cuda()
    one = _capi.Upscaler(ctx, sr, lr_shape, None, True, True, None, 1.0)

    def fresh_result(x):
        fresh = _capi.Upscaler(ctx, sr, lr_shape, None, True, True, None, 1.0)
        want = fresh(x).cpu()
        fresh.close()
        return want

    for i, n in enumerate([4, 1, 3, 2, 4, 1]):
        x = frames[:n].roll(i, dims=1).contiguous()
        got = one(x).cpu()
        want = fresh_result(x)
        assert torch.equal(got, want), f"job {i} of {n} frames differs from a new upscaler's: {int((got != want).sum())} bytes"
    # captured: static input / output tensors, replayed with new contents, eager jobs of other sizes in between
    x_static = frames[:2].clone()
    out_static = torch.empty_like(one(x_static))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        one(x_static, out_static)
    for i in range(3):
        x = frames[i:i + 2].roll(3 * i + 1, dims=2).contiguous()
        x_static.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out_static.cpu(), fresh_result(x)), f"replay {i} differs"
        y = frames[:3 - i].contiguous()
        assert torch.equal(one(y).cpu(), fresh_result(y)), f"eager job after replay {i} differs"
    del g
    one.close()


@pytest.mark.parametrize("factor", [2, 4])
def test_fsrcnn_strip_and_band_boundaries_vs_oracle(ctx, factor):
    """Widths on either side of every strip width of the matrix-core stages (mapping: 56 interior columns per workgroup in fp16 mode, 40 in
    fp32-grade mode; tail: 28 | 30 per wave; head: 128), heights from one row up to a band that straddles planes, 2-5 planes: both modes
    against the fp32 CPU forward (fp32-grade at the network tolerance, fp16 mode by PSNR)."""
    table = W.fsrcnn_table(seed=11 + factor)
    m32 = factory.build_model_fsrcnn(ctx, factor=factor, weights=table)
    m16 = factory.build_model_fsrcnn(ctx, factor=factor, weights=table, dtype="f16")
    widths = [1, 2, 27, 28, 29, 30, 31, 39, 40, 41, 55, 56, 57, 111, 112, 113, 127, 128, 129]
    heights = [1, 2, 3, 17, 33]
    worst = 99.0
    for i, wd in enumerate(widths):
        ht = heights[i % len(heights)]
        planes = 2 + i % 4
        x = torch.rand(planes, 1, ht, wd, generator=torch.Generator().manual_seed(1000 * ht + wd))
        with torch.no_grad():
            want = onets.fsrcnn(x, table, factor)
        got32 = m32(x.cuda()).cpu()
        assert_close(got32, want, what=f"fsrcnn x{factor} fp32-grade {planes}x{ht}x{wd}")
        got16 = m16(x.cuda()).float().cpu()
        assert torch.isfinite(got16).all()
        peak = float(want.abs().max())
        p = psnr(got16 / peak, want / peak)
        worst = min(worst, p)
        assert p > 60.0, f"fsrcnn x{factor} fp16 mode {planes}x{ht}x{wd}: PSNR {p:.1f} dB"
    record_measured(f"fsrcnn_boundaries_x{factor}", worst_psnr_db_f16=worst, cases=len(widths))


def test_fsrcnn_tall_bands_same_bytes_as_whole_bands_per_plane():
    """Round 6's grids (fp16 mapping stage, both matrix-core tails: the planes stacked into one tall image, cut into as many bands as fill
    the chip's workgroup slots, a band straddling plane boundaries marched in segments) against the classic ones (SS4K_MH_NO_TALL,
    SS4K_TAIL_NO_TALL: whole bands per plane), each in a child process: SHA-256 of every output tensor, shapes with one band over seven
    planes up to configs[1]'s 12 x 720 x 1280, both modes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # (third run: the fp16 mapping stage with two 64-column units per workgroup strip - a kept A/B build of the DEV library, classic bands)
    from sharkshark4k_amd import build as B
    assert os.path.exists(B.LIB_DEV), "libss4k_hip_dev.so was not built (__graft_entry__.build())"
    for switches in ({}, {"SS4K_MH_NO_TALL": "1", "SS4K_TAIL_NO_TALL": "1"}, {"SS4K_LIB": B.LIB_DEV, "SS4K_MH_NU": "2"}):
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "drive_fs_bands.py")], cwd=root, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **switches))
        assert r.returncode == 0 and "FS BANDS DONE" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("x")])
    assert len(outs[0]) == 16 and outs[0] == outs[1], [(a, b) for a, b in zip(outs[0], outs[1]) if a != b]
    assert outs[0] == outs[2], [(a, b) for a, b in zip(outs[0], outs[2]) if a != b]


# ------------------------------------------------------------------------------ fp16 HR tensor on the batched service path
@pytest.mark.parametrize("out_shape,lr_shape,n", [((144, 256), (72, 128), 2), (None, (72, 128), 1), ((180, 250), (90, 125), 3)])
def test_srvgg_f16_half_hr_tensor_vs_fp32_hr_tensor(ctx, out_shape, lr_shape, n):
    """An fp16 SRVGG writes its x4 output tensor as fp16 for the service's fused tail (statistics ride along in fp32, area map,
    normalise + colour match + clamp in place, bicubic): against the same model with SS4K_MODEL_HR_F32 the uint8 frames may
    differ by 1 LSB where a value sits within fp16's 2^-12 of a truncation boundary.  Covers the 2:1 bicubic fast path, no
    resize, and a ragged size on the generic kernels."""
    t = W.dni_blend(W.srvgg_table(3, num_conv=4), W.srvgg_table(4, num_conv=4), 0.5)
    flat = W.flatten(t, W.srvgg_keys(4))
    frames = torch.from_numpy(smooth_u8(31, (n, lr_shape[0], lr_shape[1], 3))).cuda()
    outs = []
    for fl in (_capi.MODEL_HR_F32, 0):
        sr = _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=4, num_feat=64, num_block=4, flags=fl), flat)
        up = _capi.Upscaler(ctx, sr, lr_shape, out_shape, True, False, None, 1.0)
        outs.append(up(frames).cpu().to(torch.int16))
    d = (outs[0] - outs[1]).abs()
    frac = float((d != 0).float().mean())
    record_measured(f"srvgg_half_hr_{lr_shape[0]}x{lr_shape[1]}_n{n}", max_lsb=int(d.max()), frac_differing=frac)
    print(f"fp16 vs fp32 HR tensor {lr_shape} -> {out_shape}: max {int(d.max())} LSB, {100 * frac:.2f} % of bytes differ")
    assert int(d.max()) <= 1 and frac < 0.04   # measured 2.2-2.6 % of the bytes


def test_hip_service_fsrcnn_f16_in_process(ctx):
    """HipUpscalerService(fsrcnn_dtype='f16'): the worker body in-process on the fp16 FSRCNN mode, against the oracle service
    (uint8 frames: the fp16 mode's own error is ~60 dB down, a 1 LSB flip near truncation boundaries)."""
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    table = _t91(2)
    svc = HipUpscalerService(device=0, denoising=False, upscaler_model="fsrcnn", scale=2, lr_shape=(90, 124), weights={"sr": table},
                             fsrcnn_dtype="f16")
    svc.proc_init()
    frames = torch.from_numpy(smooth_u8(5, (2, 90, 124, 3)))
    got = svc.upscale(frames.cuda()).cpu()
    osv = osvc.OracleUpscaler(lambda x: onets.fsrcnn(x, table, 2), upscaler_model="fsrcnn", lr_shape=(90, 124))
    want = osv.upscale(frames)
    d = (got.to(torch.int16) - want.to(torch.int16)).abs()
    frac = float((d != 0).float().mean())
    record_measured("svc_fsrcnn_f16_90x124", max_lsb=int(d.max()), frac_differing=frac)
    assert got.shape == want.shape and int(d.max()) <= 1 and frac < 0.06, (int(d.max()), frac)   # measured 3.8-4.4 %
