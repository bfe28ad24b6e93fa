"""GPU: the reference's SHIPPED DEFAULT network at its full depth - SRVGGNetCompact num_feat 64, num_conv 32, x4
(``realesr-general-x4v3``: ``src/upscale/model/realesrgan/factory.py:18-82,88,132-138``, DNI blend ``:152-157``) - the net behind the only
published reference figure (README: 24 fps) and behind ``bench.py``'s ``also.srvgg`` line.

Pinned: ``tests/golden/srvgg_f64_c32_x4*.npz`` and ``svc_multi_srvgg64x32_x4_area_bicubic.npz`` hold what the REFERENCE's own
``SRVGGNetCompact`` / ``FsrcnnUpscalerService`` produced at this depth (``tests/golden/make_golden.py``); the fp32 HIP path is compared with
them in ``test_gpu_parity.py`` (``test_srvgg_golden_fp32``, ``test_service_golden_u8``, the fused-tail test) like every other golden.  Here:

* fp32 against the oracle on ragged crops (partly filled tiles on every edge, one pixel wide, two frames) at rtol 1e-3 / atol 1e-4;
* the fp16 production path on a full 720p frame through ``ss4k_upscale_frames`` - x4 network, statistics + local colour match, bicubic to
  1440p - against the oracle service: PSNR and max LSB, recorded next to the asserted thresholds;
* the same with a DNI blend whose PReLU slopes leave [0, 1] (the ``max(t, t s)`` epilogue needs every slope of a layer <= 1 - checked on
  the host when the model is built; the other layers take the select form), fp16, on both builds of the 64-cout tile.
"""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from oracle import service as osvc
from tests.conftest import load_golden
from tests.helpers import assert_close, psnr, record_measured, smooth_u8, srvgg_full_table

pytestmark = pytest.mark.gpu
KEYS = W.srvgg_keys(32)


def _threads():
    import os
    torch.set_num_threads(min(16, os.cpu_count() or 1))


def _model(ctx, table, dtype, flags=0):
    return _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, dtype, scale=4, num_feat=64, num_block=32, flags=flags), W.flatten(table, KEYS))


@pytest.mark.parametrize("shape", [(1, 3, 37, 70), (2, 3, 33, 47), (1, 3, 5, 7), (1, 3, 64, 1)])
@pytest.mark.parametrize("seeds", [(14, None), (15, 16)])
def test_srvgg32_fp32_vs_oracle_ragged(ctx, seeds, shape):
    """All 34 convs in fp32 against the oracle, north_star's tolerance taken literally (the output is image-range: input + a small residue)."""
    _threads()
    table = srvgg_full_table(*seeds)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2] * 131 + shape[3]))
    with torch.no_grad():
        want = onets.srvgg(x, table, 32, 4)
    got = _model(ctx, table, _capi.F32)(x.cuda())
    assert_close(got, want, rtol=1e-3, atol=1e-4, what=f"srvgg 64x32 x4 fp32 {shape} seeds {seeds}")
    record_measured(f"srvgg32_fp32_{shape[0]}x{shape[2]}x{shape[3]}_{'dni_wild' if seeds[1] else 'plain'}",
                    max_abs_err=float((got.cpu() - want).abs().max()), out_peak=float(want.abs().max()), asserted="rtol 1e-3, atol 1e-4")


def test_srvgg32_wild_slopes_reach_both_epilogue_forms():
    """The DNI table of these tests really has layers on both sides of the host check (every slope <= 1 or not), and slopes below zero."""
    t = srvgg_full_table(15, 16)
    le1 = [bool((np.asarray(t[f"body.{2 * i + 1}.weight"]) <= 1.0).all()) for i in range(33)]
    assert 10 <= sum(le1) <= 23, le1
    assert min(float(np.asarray(t[f"body.{2 * i + 1}.weight"]).min()) for i in range(33)) < -0.4


# asserted at measured - 2 dB / + 1 LSB; measured values: profiles/earlier/r05/r05_parity_measured.json
S32_PSNR_DB, S32_MAX_LSB = 61.8, 2     # measured 63.89 dB, 1 LSB (2.7 % of the bytes differ)
S32W_PSNR_DB, S32W_MAX_LSB = 61.5, 2   # measured 63.56 dB, 1 LSB (2.9 %)


@pytest.mark.parametrize("seeds,tag,bar", [((14, None), "plain", (S32_PSNR_DB, S32_MAX_LSB)), ((15, 16), "dni_wild", (S32W_PSNR_DB, S32W_MAX_LSB))])
def test_srvgg32_720p_fp16_service_vs_oracle(ctx, seeds, tag, bar):
    """The reference's default deployment (pipeline.py:41-50: x4 network on 720p, output_shape 1440p) in the production dtype, exactly as
    bench.py's `srvgg` workload runs it, against the oracle service on the host (one 2.2 TFLOP forward)."""
    _threads()
    table = srvgg_full_table(*seeds)
    up = _capi.Upscaler(ctx, _model(ctx, table, _capi.F16), (720, 1280), (1440, 2560), True, False, None, 1.0)
    frames = torch.from_numpy(smooth_u8(321, (1, 720, 1280, 3)))
    got = up(frames.cuda()).cpu()
    osv = osvc.OracleUpscaler(lambda x: onets.srvgg(x, table, 32, 4), upscaler_model="realesrgan", lr_shape=(720, 1280), output_shape=(1440, 2560))
    want = osv.upscale(frames)
    assert got.shape == (1, 1440, 2560, 3) and got.dtype == torch.uint8
    d = (got.int() - want.int()).abs()
    p = psnr(got.float(), want.float(), peak=255.0)
    print(f"srvgg 64x32 x4 ({tag}) 720p fp16 service vs oracle: PSNR {p:.2f} dB, max |delta| {int(d.max())} LSB, {float((d > 0).float().mean()):.4%} bytes differ")
    record_measured(f"srvgg32_{tag}_720p_fp16_service", psnr_db=p, max_lsb=int(d.max()), bytes_differ=float((d > 0).float().mean()),
                    asserted=f"PSNR >= {bar[0]} dB, max <= {bar[1]} LSB")
    assert p >= bar[0], f"PSNR {p:.2f} dB"
    assert int(d.max()) <= bar[1], f"max |delta| {int(d.max())} LSB"
    # frames are independent: a 4-frame job (two launch chains) gives this frame the same bytes
    four = torch.cat([frames, torch.from_numpy(smooth_u8(322, (3, 720, 1280, 3)))]).cuda()
    assert torch.equal(up(four)[0].cpu(), got[0])


@pytest.mark.parametrize("flags,what", [(0, "16x16x32 tile (default)"), (_capi.MODEL_NO_W16, "32x32x16 tile")])
def test_srvgg32_fp16_network_vs_reference_golden(ctx, flags, what):
    """The fp16 network's float output against what the REFERENCE's SRVGGNetCompact produced at full depth (fp32), both weight tables, on
    both builds of the 64-cout tile: the two must be equally far from it."""
    # measured (peak 1.0; the output is the input plus a small residue, so the fp32 base dominates): 102.4 dB / 115.9 dB on either tile
    for name, seeds, bar in (("srvgg_f64_c32_x4", (14, None), 100.0), ("srvgg_f64_c32_x4_dni_wild", (15, 16), 113.5)):
        g = load_golden(name)
        y = _model(ctx, srvgg_full_table(*seeds), _capi.F16, flags)(torch.from_numpy(g["x"]).cuda()).cpu()
        p = psnr(y, torch.from_numpy(g["y"]))
        record_measured(f"{name}_fp16_{'w16' if flags == 0 else 'wide'}", psnr_db=p, max_abs_err=float((y - torch.from_numpy(g['y'])).abs().max()),
                        asserted=f"PSNR >= {bar} dB (peak 1.0)")
        print(f"{name} fp16 on the {what}: {p:.1f} dB vs the reference's fp32 output")
        assert p >= bar, (name, what, p)
