"""GPU: the headline path (BASELINE configs[2], [3], [4]) at full network depth against the oracle.

RRDBNet's oracle restates the published BasicSR architecture (parity unpinned: basicsr is not
vendored by the reference, see oracle/__init__.py); everything else on these paths (BSVD, service
glue) is pinned by the reference-generated fixtures in tests/golden.

* fp32 HIP path vs oracle: rtol 1e-3 / atol 1e-4 (north_star) with all 23 RRDB blocks, so the
  buffer rotation and the in-place ``res2`` aliasing of ``Model::forward`` run over all 69 dense blocks.
* fp16 production path at full size through ``ss4k_upscale_frames`` vs the oracle *service*:
  PSNR and max |delta| of the uint8 frames.
* configs[3] as north_star names it: BSVD-32 + RRDBNet x2 through the per-frame path
  (``fsrcnn_upscaler.py:262,269-271,292-299``), first and later job.
"""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.upscale import model as factory
from oracle import nets as onets
from oracle import service as osvc
from tests.helpers import assert_close, assert_u8_close, psnr, record_measured, smooth_u8

pytestmark = pytest.mark.gpu


def _cpu_threads():
    # many-core hosts oversubscribe these small convs badly; 16 threads is the sweet spot measured in round 1
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))


# ------------------------------------------------------------------------------ (a) 23 blocks, fp32, vs oracle
@pytest.mark.parametrize("scale,shape", [(2, (1, 3, 96, 160)), (2, (2, 3, 64, 136)), (4, (1, 3, 40, 72)), (1, (1, 3, 128, 192))])
def test_rrdbnet_23_blocks_fp32_vs_oracle(ctx, scale, shape):
    _cpu_threads()
    table = W.rrdbnet_table(31 + scale, scale=scale)  # 23 blocks, 64 features, growth 32
    m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F32, scale=scale), W.flatten(table, W.rrdbnet_keys(23)))
    x = torch.from_numpy(smooth_u8(3 + scale, (shape[0], shape[2], shape[3], 3))).permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        want = onets.rrdbnet(x, table, scale, 23)
    got = m(x.cuda())
    # north_star's atol = 1e-4 is quoted for image-range outputs ([0,1]); this random-weight network's
    # output peaks near 10, so the absolute term is scaled by the output's peak (rtol stays 1e-3)
    peak = max(1.0, float(want.abs().max()))
    assert_close(got, want, rtol=1e-3, atol=1e-4 * peak, what=f"rrdbnet x{scale} 23 blocks {shape}")
    err = float((got.cpu() - want).abs().max())
    print(f"rrdbnet x{scale} 23 blocks {shape}: output peak {peak:.3g}, max |err| {err:.3g} = {err / peak:.2e} of peak")
    assert err / peak < 1e-4


def _image_range(table, gain=0.01):
    """A trained network's output is image-range; the 0.1-scaled Kaiming tables peak near 10.  Same body, conv_last scaled
    and biased so that the output sits in [0,1] - the range north_star's atol = 1e-4 is quoted for."""
    t = dict(table)
    t["conv_last.weight"] = table["conv_last.weight"] * np.float32(gain)
    t["conv_last.bias"] = np.full_like(table["conv_last.bias"], 0.5)
    return t


@pytest.mark.parametrize("scale,shape", [(2, (1, 3, 96, 160)), (4, (1, 3, 40, 72)), (1, (1, 3, 128, 192))])
def test_rrdbnet_23_blocks_fp32_image_range_literal_tolerance(ctx, scale, shape):
    """All 23 blocks, fp32, output in image range: north_star's tolerance taken literally (rtol 1e-3, atol 1e-4)."""
    _cpu_threads()
    raw = W.rrdbnet_table(31 + scale, scale=scale)
    x = torch.from_numpy(smooth_u8(3 + scale, (shape[0], shape[2], shape[3], 3))).permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        # conv_last is linear: one oracle pass finds the gain that puts this table's output at 0.5 +- 0.45
        dev0 = float((onets.rrdbnet(x, _image_range(raw), scale, 23) - 0.5).abs().max())
        table = _image_range(raw, gain=0.01 * 0.45 / dev0)
        want = onets.rrdbnet(x, table, scale, 23)
    m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F32, scale=scale), W.flatten(table, W.rrdbnet_keys(23)))
    assert 0.0 < float(want.min()) and float(want.max()) < 1.0, (float(want.min()), float(want.max()))
    got = m(x.cuda())
    assert_close(got, want, rtol=1e-3, atol=1e-4, what=f"rrdbnet x{scale} 23 blocks, image-range output")
    err = float((got.cpu() - want).abs().max())
    record_measured(f"rrdbnet_x{scale}_23blocks_fp32_image_range", max_abs_err=err, out_min=float(want.min()), out_max=float(want.max()),
                    asserted="rtol 1e-3, atol 1e-4")


def test_rrdbnet_23_blocks_block_slip_is_detected(ctx):
    """The comparison above is sensitive to a single late block: perturbing body.22.rdb3.conv5 of the
    HIP model's weights by 50 % moves more than 1 % of the outputs out of the tolerance."""
    _cpu_threads()
    table = W.rrdbnet_table(33, scale=2)
    x = torch.from_numpy(smooth_u8(5, (1, 64, 96, 3))).permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        want = onets.rrdbnet(x, table, 2, 23)
    bad = dict(table)
    bad["body.22.rdb3.conv5.weight"] = table["body.22.rdb3.conv5.weight"] * 1.5
    m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F32, scale=2), W.flatten(bad, W.rrdbnet_keys(23)))
    err = (m(x.cuda()).cpu() - want).abs()
    peak = max(1.0, float(want.abs().max()))
    assert float((err > 1e-4 * peak + 1e-3 * want.abs()).float().mean()) > 0.01


# ------------------------------------------------------------------------------ (b) configs[2] at full size, fp16, service path
def test_config2_rrdbnet_x2_720p_fp16_service_vs_oracle(ctx):
    """BASELINE configs[2] exactly as bench.py runs it: 23-block RRDBNet x2, 720p -> 1440p, fp16 storage,
    through ss4k_upscale_frames (batched path: stats match, local colour match, truncation), against
    the oracle service run on the host (one 8.3 TFLOP forward)."""
    _cpu_threads()
    table = W.rrdbnet_table(0, scale=2)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(table, W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, (720, 1280), None, True, False, None, 1.0)
    frames = torch.from_numpy(smooth_u8(123, (1, 720, 1280, 3)))
    got = up(frames.cuda()).cpu()
    osv = osvc.OracleUpscaler(lambda x: onets.rrdbnet(x, table, 2, 23), upscaler_model="realesrgan", lr_shape=(720, 1280))
    want = osv.upscale(frames)
    assert got.shape == (1, 1440, 2560, 3) and got.dtype == torch.uint8
    d = (got.int() - want.int()).abs()
    p = psnr(got.float(), want.float(), peak=255.0)
    print(f"configs[2] fp16 vs oracle: PSNR {p:.2f} dB, max |delta| {int(d.max())} LSB, {float((d > 0).float().mean()):.4%} bytes differ")
    record_measured("config2_rrdbnet_x2_720p_fp16_service", psnr_db=p, max_lsb=int(d.max()), bytes_differ=float((d > 0).float().mean()),
                    bytes_2lsb=int((d >= 2).sum()), asserted="PSNR >= 55.5 dB, max <= 2 LSB, at most 16 bytes at 2 LSB")
    # measured 57.50 dB / 2 LSB - the 2 in ONE byte of 11 059 200 (profiles/earlier/r05/r05_parity_measured.json: bytes_2lsb = 1); with either of the
    # two not-bit-identical route choices pinned off (NO_W16, NO_UPS_PRESUM) the worst byte is 1 LSB, with both off it is 2 again: a byte
    # on a rounding edge that any change of summation order moves, not the cost of one route.  Asserted at measured - 2 dB, the measured
    # worst byte, and a count: a handful of rounding-edge bytes may reach 2 LSB, thousands of them (or one byte at 3) is a regression.
    assert p >= 55.5, f"PSNR {p:.2f} dB"
    assert int(d.max()) <= 2, f"max |delta| {int(d.max())} LSB"
    assert int((d >= 2).sum()) <= 16, f"{int((d >= 2).sum())} bytes at 2 LSB"
    # which route costs what against the oracle (round 4 moved the worst byte from 1 to 2 LSB): each non-bit-identical choice pinned off in turn
    for tag, fl in (("no_ups_presum", _capi.MODEL_NO_UPS_PRESUM), ("no_w16", _capi.MODEL_NO_W16), ("no_w16_no_ups_presum", _capi.MODEL_NO_W16 | _capi.MODEL_NO_UPS_PRESUM)):
        sr_r = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, flags=fl), W.flatten(table, W.rrdbnet_keys(23)))
        g_r = _capi.Upscaler(ctx, sr_r, (720, 1280), None, True, False, None, 1.0)(frames.cuda()).cpu()
        d_r = (g_r.int() - want.int()).abs()
        record_measured(f"config2_rrdbnet_x2_720p_fp16_service_{tag}", psnr_db=psnr(g_r.float(), want.float(), peak=255.0), max_lsb=int(d_r.max()),
                        bytes_differ=float((d_r > 0).float().mean()), bytes_2lsb=int((d_r >= 2).sum()))
        del sr_r
    # frames are independent: a 4-frame job gives every frame the result of a 1-frame job, bit for bit - every layer runs on the same
    # kernel whatever the job size, on the default route and on the 32x32x16 build of the 64-cout tile (SS4K_MODEL_NO_W16)
    four = torch.cat([frames, torch.from_numpy(smooth_u8(124, (3, 720, 1280, 3)))]).cuda()
    assert torch.equal(up(four)[0].cpu(), got[0])
    sr_p = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, flags=_capi.MODEL_NO_W16), W.flatten(table, W.rrdbnet_keys(23)))
    up_p = _capi.Upscaler(ctx, sr_p, (720, 1280), None, True, False, None, 1.0)
    assert torch.equal(up_p(four)[0].cpu(), up_p(frames.cuda())[0].cpu())


# ------------------------------------------------------------------------------ (c) configs[3]: BSVD + RRDBNet, per-frame path
def _pipeline(ctx, dtype, lr_shape, sr_table, bs_table, num_block, out_shape=None, rate=1.0):
    d = _capi.F32 if dtype == "f32" else _capi.F16
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, d, scale=2, num_block=num_block), W.flatten(sr_table, W.rrdbnet_keys(num_block)))
    dn = factory.build_denoise_model(ctx, weights=bs_table, dtype=dtype)
    up = _capi.Upscaler(ctx, sr, lr_shape, out_shape, True, True, dn, rate)
    osv = osvc.OracleUpscaler(lambda x: onets.rrdbnet(x, sr_table, 2, num_block), denoising=True, denoise_rate=rate,
                              upscaler_model="realesrgan", denoise_model=lambda x: onets.bsvd_f1(x, bs_table),
                              output_shape=out_shape, single_mode=True, lr_shape=lr_shape)
    return up, osv, (sr, dn)


@pytest.mark.parametrize("rate,out_shape,in_hw", [(1.0, None, (48, 80)), (0.3, (100, 170), (61, 99))])
def test_config3_bsvd_rrdbnet_fp32_vs_oracle_small(ctx, rate, out_shape, in_hw):
    """denoise + realesrgan branch of upscale_single (fsrcnn_upscaler.py:292-295) with the real SR
    architecture of configs[3] (23-block RRDBNet x2), fp32: u8 frames <= 1 LSB and float taps within
    the parity tolerance; first job (noise map 0.05) and later job (0.1 * denoise_rate)."""
    _cpu_threads()
    sr_table, bs_table = W.rrdbnet_table(41, scale=2), W.bsvd_table(seed=21)
    up, osv, keep = _pipeline(ctx, "f32", (48, 80), sr_table, bs_table, 23, out_shape, rate)
    frames = torch.from_numpy(smooth_u8(61, (2, in_hw[0], in_hw[1], 3)))
    up.enable_taps(True)
    for job in range(2):
        taps = [dict() for _ in range(2)]
        want = torch.stack([osv.upscale_single(frames[i], taps[i]) for i in range(2)])
        got = up(frames.cuda())
        assert_u8_close(got, want, what=f"configs[3] fp32 job {job}")
        for which, key in ((0, "lr"), (1, "model"), (4, "final")):
            w = torch.stack([t[key] for t in taps])
            if key != "lr":
                w = w[:, :, 0]
            assert_close(up.read_tap(which), w, what=f"configs[3] fp32 job {job} tap {key}")


# measured (profiles/earlier/r05/r05_parity_measured.json) + 1 LSB / - 2 dB
C3_PSNR_DB, C3_MAX_LSB = 55.9, 2   # measured 57.93 dB, 1 LSB (both jobs)


def test_config3_bsvd_rrdbnet_720p_fp16_vs_oracle(ctx):
    """BASELINE configs[3] at full size in the production dtype: BSVD-32 + 23-block RRDBNet x2 on a
    720p frame through the per-frame path; first job and later job against the oracle service."""
    _cpu_threads()
    sr_table, bs_table = dict(W.rrdbnet_table(0, scale=2)), W.bsvd_table(seed=0)
    # The random-weight network's raw output peaks near 50; this path clamps it to [0,1] BEFORE the
    # statistics match (fsrcnn_upscaler.py:298-299), so fp16 storage noise of +-0.03 at that magnitude
    # would show up as tens of LSB on the few unsaturated pixels.  A trained net's output is image-range:
    # give the synthetic one an image-range output too (same table for the oracle and the HIP path).
    sr_table = _image_range(sr_table)
    up, osv, keep = _pipeline(ctx, "f16", (720, 1280), sr_table, bs_table, 23)
    f0 = torch.from_numpy(smooth_u8(123, (1, 720, 1280, 3)))
    for job in range(2):
        got = up(f0.cuda()).cpu()
        want = osv.upscale(f0)
        assert got.shape == (1, 1440, 2560, 3)
        d = (got.int() - want.int()).abs()
        p = psnr(got.float(), want.float(), peak=255.0)
        print(f"configs[3] fp16 job {job}: PSNR {p:.2f} dB, max |delta| {int(d.max())} LSB")
        record_measured(f"config3_bsvd_rrdbnet_720p_fp16_job{job}", psnr_db=p, max_lsb=int(d.max()), bytes_differ=float((d > 0).float().mean()),
                        saturated=float(((want == 0) | (want == 255)).float().mean()), asserted=f"PSNR >= {C3_PSNR_DB} dB, max <= {C3_MAX_LSB} LSB")
        assert p >= C3_PSNR_DB, f"job {job}: PSNR {p:.2f} dB"
        assert int(d.max()) <= C3_MAX_LSB, f"job {job}: max |delta| {int(d.max())}"
    # the first-frame noise level differs from later frames' (0.05 vs 0.1): the two jobs must not be identical
    up.reset()
    first = up(f0.cuda())
    later = up(f0.cuda())
    assert not torch.equal(first, later)


# ------------------------------------------------------------------------------ (d) configs[4] per GPU at full depth
def test_config4_rrdbnet_x4_1080p_23_blocks(ctx):
    """BASELINE configs[4] on one GPU at full depth: 23-block RRDBNet x4, 1080p -> 4320x7680, bicubic
    to 2160x3840 through the batched service path.  Size-independent checks: batch splitting does not
    change a frame, statistics follow the input (stats match), and a far-corner crop of the network
    output equals the network run on the matching input crop (shift invariance, interior only)."""
    table = W.rrdbnet_table(2, scale=4)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=4), W.flatten(table, W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, (1080, 1920), (2160, 3840), True, False, None, 1.0)
    frames = torch.from_numpy(smooth_u8(9, (3, 1080, 1920, 3))).cuda()
    out = up(frames)
    assert out.shape == (3, 2160, 3840, 3) and out.dtype == torch.uint8
    assert torch.equal(up(frames[2:3])[0], out[2])
    fi, fo = frames.float(), out.float()
    assert abs(float(fi.mean()) - float(fo.mean())) < 2.0
    assert abs(float(fi.std()) - float(fo.std())) < 6.0


def test_config4_rrdbnet_x4_23_blocks_fp16_vs_oracle_crop(ctx):
    """The production dtype of configs[4] against the oracle at full depth: 23-block RRDBNet x4, fp16 storage, on a
    270x480 crop (a quarter of the 1080p frame each way; output 1080x1920) through ss4k_upscale_frames with the config's
    bicubic to half the network's output size, image-range conv_last as in the configs[3] test."""
    _cpu_threads()
    table = _image_range(W.rrdbnet_table(2, scale=4))
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=4), W.flatten(table, W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, (270, 480), (540, 960), True, False, None, 1.0)
    frames = torch.from_numpy(smooth_u8(19, (1, 270, 480, 3)))
    got = up(frames.cuda()).cpu()
    osv = osvc.OracleUpscaler(lambda x: onets.rrdbnet(x, table, 4, 23), upscaler_model="realesrgan", lr_shape=(270, 480),
                              output_shape=(540, 960))
    want = osv.upscale(frames)
    assert got.shape == (1, 540, 960, 3) and got.dtype == torch.uint8
    d = (got.int() - want.int()).abs()
    p = psnr(got.float(), want.float(), peak=255.0)
    record_measured("config4_rrdbnet_x4_23blocks_fp16_270x480_crop", psnr_db=p, max_lsb=int(d.max()), bytes_differ=float((d > 0).float().mean()),
                    asserted=f"PSNR >= {C4_PSNR_DB} dB, max <= {C4_MAX_LSB} LSB")
    print(f"configs[4] fp16 x4 23 blocks vs oracle (270x480 crop): PSNR {p:.2f} dB, max |delta| {int(d.max())} LSB")
    assert p >= C4_PSNR_DB, f"PSNR {p:.2f} dB"
    assert int(d.max()) <= C4_MAX_LSB, f"max |delta| {int(d.max())} LSB"
    # the network output itself (float, before the service glue) against the oracle
    x = frames.permute(0, 3, 1, 2).float().div(255.0)
    with torch.no_grad():
        wy = onets.rrdbnet(x, table, 4, 23)
    gy = sr(x.cuda()).cpu()
    pn = psnr(gy, wy, peak=1.0)
    record_measured("config4_rrdbnet_x4_23blocks_fp16_network_output", psnr_db=pn, max_abs_err=float((gy - wy).abs().max()), asserted="PSNR >= 50 dB (peak 1.0)")
    assert pn >= 50.0, f"network output PSNR {pn:.2f} dB"


def test_job_whose_plane_exceeds_4gb_is_routed_as_a_whole(ctx):
    """The 16x16x32 / wide / narrow tile kernels keep 32-bit byte offsets inside a plane (< 4 GB of fp16 records).  Five 1080p frames through an
    x4 RRDBNet make the tail's planes 5.3 GB; the job runs as two launch chains (frames 0-1, frames 2-4), and lane 1 alone is under the limit
    (3 frames = 3.2 GB) while its pixel offsets - counted from frame 0 - are not.  The kernel choice must therefore follow the JOB's span:
    five identical frames in, five identical frames out (a wrapped offset reads another frame's rows)."""
    t = W.rrdbnet_table(6, scale=4, num_block=1)
    m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=4, num_block=1, flags=_capi.MODEL_TWO_CHAINS), W.flatten(t, W.rrdbnet_keys(1)))
    one = torch.rand(1, 3, 1080, 1920, generator=torch.Generator().manual_seed(4)).cuda()
    y = m(one.expand(5, -1, -1, -1).contiguous())
    assert y.shape == (5, 3, 4320, 7680) and torch.isfinite(y).all()
    for k in range(1, 5):
        assert torch.equal(y[k], y[0]), f"frame {k} of the five-frame job differs from frame 0"
    del y
    torch.cuda.empty_cache()


C4_PSNR_DB, C4_MAX_LSB = 55.5, 2   # measured 57.55 dB, 1 LSB (profiles/earlier/r05/r05_parity_measured.json)


# ------------------------------------------------------------------------------ (f) fp16 storage at realistic activation ranges
@pytest.mark.parametrize("nb", [6, 23])
def test_rrdbnet_fp16_full_gain_weights(ctx, nb):
    """Trained RealESRGAN checkpoints have far larger activations than the 0.1-scaled Kaiming init of
    the synthetic tables.  Gain-1.0 Kaiming RDB weights (no 0.1 damping) and full-range inputs: the
    fp16-storage path must stay finite and close to the fp32 oracle; the PSNR is reported.  At 23 blocks the trunk
    grows by the RRDB residual alone to a peak of ~220 (1.2 per block), the depth a trained checkpoint runs at."""
    _cpu_threads()
    import math
    table = W.rrdbnet_table(51, scale=2, num_block=nb)
    for k in list(table):
        if ".rdb" in k and k.endswith(".weight"):
            cout, cin = table[k].shape[:2]
            table[k] = (W._normal(51, k + ".full", table[k].shape, math.sqrt(2.0 / (cin * 9)))).astype(np.float32)
    x = torch.from_numpy(np.random.default_rng(5).integers(0, 256, (1, 3, 96, 128)).astype(np.float32) / 255.0)
    with torch.no_grad():
        want = onets.rrdbnet(x, table, 2, nb)
    m16 = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=nb), W.flatten(table, W.rrdbnet_keys(nb)))
    m32 = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F32, scale=2, num_block=nb), W.flatten(table, W.rrdbnet_keys(nb)))
    y16, y32 = m16(x.cuda()).cpu(), m32(x.cuda()).cpu()
    assert torch.isfinite(y16).all(), "fp16 storage overflowed"
    assert_close(y32, want, rtol=1e-3, atol=1e-4 * max(1.0, float(want.abs().max())), what="fp32 path, full-gain weights")
    peak = float(want.abs().max())
    p = psnr(y16, want, peak=peak)
    print(f"fp16 vs oracle with gain-1.0 RDB weights, {nb} blocks: output peak {peak:.3g}, PSNR {p:.1f} dB")
    bar = {6: 72.0, 23: 67.0}[nb]   # measured 74.4 dB / 69.1 dB (profiles/earlier/r05/r05_parity_measured.json), asserted at - 2 dB
    record_measured(f"rrdbnet_fp16_full_gain_{nb}blocks", psnr_db=p, out_peak=peak, asserted=f"PSNR > {bar} dB (peak = output peak)")
    assert p > bar, f"PSNR {p:.1f} dB at output peak {peak:.3g}"


# ------------------------------------------------------------------------------ frame lanes (two concurrent launch chains)
@pytest.mark.parametrize("n", [4, 3, 5, 2])
@pytest.mark.parametrize("kind", ["rrdbnet", "bsvd", "srvgg"])
def test_frame_lanes_bit_identical(ctx, monkeypatch, kind, n):
    """An fp16 batch of two or more frames may go through the conv layers as two concurrent launch chains (csrc/models.h,
    frame lanes); an odd batch splits unevenly (3 = 1 + 2, 5 = 2 + 3), so the two chains run different grid sizes.
    One chain (SS4K_LANES=1), two chains (=2) and the measured choice (unset: calls 0-5 run two / one / two / one / one / two chains, then the
    faster) must give bit-identical tensors, call after call, for every conv network."""
    def build():
        if kind == "rrdbnet":
            t = W.rrdbnet_table(5, scale=2, num_block=2)
            return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2), W.flatten(t, W.rrdbnet_keys(2)))
        if kind == "bsvd":
            return _capi.Model(ctx, _capi.make_desc(_capi.BSVD, _capi.F16, scale=1), W.flatten(W.bsvd_table(3), W.bsvd_keys()))
        t = W.srvgg_table(2, num_feat=64, num_conv=4)
        return _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F16, scale=4, num_feat=64, num_block=4), W.flatten(t, W.srvgg_keys(4)))
    cin = 4 if kind == "bsvd" else 3
    x = torch.rand(n, cin, 72, 104, generator=torch.Generator().manual_seed(11)).cuda()
    outs = {}
    for mode in ("1", "2", None):
        if mode is None:
            monkeypatch.delenv("SS4K_LANES", raising=False)
        else:
            monkeypatch.setenv("SS4K_LANES", mode)
        m = build()
        ys = [m(x).clone() for _ in range(9)]   # the measured choice alternates modes over its first six calls, then settles
        torch.cuda.synchronize()
        for y in ys[1:]:
            assert torch.equal(ys[0], y), f"{kind}: lanes mode {mode}: output changed between calls"
        outs[mode] = ys[0]
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["2"]), f"{kind}: two launch chains changed the result"
    assert torch.equal(outs["1"], outs[None]), f"{kind}: measured lane choice changed the result"
    # an odd batch always takes one chain and equals the even batch's first frames
    monkeypatch.setenv("SS4K_LANES", "2")
    m = build()
    assert torch.equal(m(x[:3]), outs["1"][:3])


def test_tile_height_builds_bit_identical(ctx):
    """The 32-cout body layers run on 16-row or 20-row tiles (csrc/conv_mfma.hip: whichever cuts the image rows with less
    waste, SS4K_MODEL_TILE_ROWS_16 / _20 force one): tiles only partition the pixels, so the network output must not change by a bit -
    including ragged heights where the last tile row of either shape is partly outside the image."""
    t = W.rrdbnet_table(8, scale=2, num_block=2)
    def build(flags):
        return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2, flags=flags), W.flatten(t, W.rrdbnet_keys(2)))
    m4, m5, ma = build(_capi.MODEL_TILE_ROWS_16), build(_capi.MODEL_TILE_ROWS_20), build(0)
    for shape in ((2, 3, 120, 136), (1, 3, 92, 200), (3, 3, 80, 72)):   # body grids 60x68, 46x100, 40x36
        x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2])).cuda()
        a, b, c = m4(x).clone(), m5(x).clone(), ma(x).clone()
        assert torch.isfinite(a).all()
        assert torch.equal(a, b) and torch.equal(a, c), f"{shape}: tile height changed the result"


def test_forward_that_throws_after_the_fork_joins_the_lane_stream(ctx, monkeypatch):
    """A forward that throws between the fork of the second launch chain and its join (here: injected at the 9th conv call,
    dev library only, every other forward) must leave no lane-1 kernels racing the next call: Model::forward's guard makes the caller's stream
    wait for the lane stream.  The following calls - same model, same buffers - are bit-identical to a model that never
    failed, and the error is reported through the C ABI."""
    import ctypes as C
    from sharkshark4k_amd import build as B
    L = _capi.load(B.LIB_DEV)
    t = W.rrdbnet_table(5, scale=2, num_block=2)
    flat = np.ascontiguousarray(W.flatten(t, W.rrdbnet_keys(2)), dtype=np.float32)
    desc = _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2, flags=_capi.MODEL_TWO_CHAINS)
    x = torch.rand(4, 3, 144, 208, generator=torch.Generator().manual_seed(3)).cuda()
    want = _capi.Model(ctx, desc, flat)(x).clone()

    hctx, hm = C.c_void_p(), C.c_void_p()
    assert L.ss4k_ctx_create(0, C.byref(hctx)) == 0
    monkeypatch.setenv("SS4K_FAIL_AT_CONV", "9")   # read when the model is built
    assert L.ss4k_model_create(hctx, C.byref(desc), flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(hm)) == 0, L.ss4k_last_error()
    monkeypatch.delenv("SS4K_FAIL_AT_CONV")
    out = torch.empty_like(want)
    st = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(4):
        # the injection hits every other forward of this model: a failed call, then a healthy one on the SAME activation buffers
        rc = L.ss4k_model_forward(hm, x.data_ptr(), out.data_ptr(), 4, 144, 208, st)
        assert rc == -22 and b"injected failure" in L.ss4k_last_error()
        out.fill_(-1.0)
        assert L.ss4k_model_forward(hm, x.data_ptr(), out.data_ptr(), 4, 144, 208, st) == 0, L.ss4k_last_error()
        torch.cuda.synchronize()
        assert torch.equal(out, want)
    L.ss4k_model_destroy(hm)
    L.ss4k_ctx_destroy(hctx)
