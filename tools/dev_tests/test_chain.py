"""GPU: the cross-layer chain kernel (csrc/conv_chain.hip) - the RRDB body of a small job as ONE persistent launch with
per-tile hand-offs between the layers - against the one-launch-per-layer path.

The chain runs every layer on the LDS-weights kernel's 32-cout tile body, so its results must be BIT-IDENTICAL to the
per-launch path with conv5's residual read from memory (SS4K_NO_RL=1; same MFMA order per output, same epilogue arithmetic); against the default
per-launch route (conv5 on the register-stationary kernel) only the order of fp32 additions inside a layer differs.
Every hand-off is exercised on reused buffers (the growth planes are rewritten every RDB, the trunk buffers rotate), so a
stale read or a too-early write anywhere in the 345-layer chain changes the output."""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from tests.helpers import psnr, smooth_u8

pytestmark = pytest.mark.gpu

# dev library only since round 5 (include/ss4k_dev.h).  NO_RS is a pseudo-flag of this file: "conv5's residual read from memory in the
# epilogue" (what the chain's tile body does) = the dev library's SS4K_NO_RL=1, read when a model is built
NO_CHAIN, CHAIN, NO_RS = 0, _capi.DEV_MODEL_CHAIN, 1 << 30


def _model(ctx, flat, scale, nb, flags):
    # every layer outside the chain (and conv5 of the reference path) on the 32x32x16 kernels the chain's tile body is bit-identical to:
    # conv_w16.hip (the default route of 64-cout layers since round 4) adds in another order
    import os
    if flags & NO_RS:
        os.environ["SS4K_NO_RL"] = "1"
    try:
        return _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=nb, flags=(flags & ~NO_RS) | _capi.MODEL_NO_W16), flat)
    finally:
        os.environ.pop("SS4K_NO_RL", None)


@pytest.mark.parametrize("scale,shape,rows", [(2, (1, 3, 144, 208), 0), (2, (2, 3, 92, 200), 16), (2, (3, 3, 80, 72), 20),
                                              (4, (1, 3, 37, 70), 0), (1, (1, 3, 128, 256), 0), (2, (1, 3, 360, 500), 0),
                                              (2, (1, 3, 16, 20), 0), (2, (3, 3, 24, 60), 16), (4, (2, 3, 9, 33), 0)])
def test_chain_bit_identical_to_launches(ctx, scale, shape, rows):
    """Forced chain (any batch) vs one launch per layer on the same tile body; ragged sizes put partly filled tiles on every
    edge, several frames put several frames' tiles into one queue, both tile heights are covered; the last three shapes are
    one, two and three tiles per frame (fewer tiles than the distance rule for the grid assumes: one workgroup does it all)."""
    tab = W.rrdbnet_table(71 + scale, scale=scale, num_block=3)
    flat = W.flatten(tab, W.rrdbnet_keys(3))
    tr = {0: 0, 16: _capi.MODEL_TILE_ROWS_16, 20: _capi.MODEL_TILE_ROWS_20}[rows]
    ref = _model(ctx, flat, scale, 3, NO_CHAIN | NO_RS | tr)
    ch = _model(ctx, flat, scale, 3, CHAIN | tr)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(shape[2])).cuda()
    want = ref(x).clone()
    assert torch.isfinite(want).all()
    for _ in range(3):
        got = ch(x)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"{shape}: chain differs from the per-launch path, max |d| {float((got - want).abs().max()):.3g}"


def test_chain_matches_oracle_like_the_default_route(ctx):
    """The chain's output stays as close to the oracle as the default multi-frame route (frame lanes, conv5 on the
    register-stationary kernel); the default route does not use the chain."""
    tab = W.rrdbnet_table(9, scale=2, num_block=4)
    flat = W.flatten(tab, W.rrdbnet_keys(4))
    x = torch.from_numpy(smooth_u8(2, (2, 96, 160, 3))).permute(0, 3, 1, 2).float().div(255.0).cuda()
    dflt, forced, ref = _model(ctx, flat, 2, 4, 0), _model(ctx, flat, 2, 4, CHAIN), _model(ctx, flat, 2, 4, NO_CHAIN | NO_RS)
    a, b, c = dflt(x[:1]).clone(), forced(x[:1]).clone(), ref(x[:1]).clone()
    assert torch.equal(b, c)
    assert torch.equal(a, _model(ctx, flat, 2, 4, NO_CHAIN)(x[:1]))
    with torch.no_grad():
        want = onets.rrdbnet(x.cpu(), tab, 2, 4)
    two = dflt(x)
    peak = float(want.abs().max())
    p1, p2 = psnr(b.cpu(), want[:1], peak=peak), psnr(two.cpu(), want, peak=peak)
    assert p1 > 55.0 and p2 > 55.0 and abs(p1 - p2) < 1.5, (p1, p2)


def test_chain_720p_23_blocks_repeatable(ctx):
    """The headline network on one 720p frame: 12 runs of the chain give 12 identical tensors, equal to the per-launch path."""
    flat = W.flatten(W.rrdbnet_table(0, scale=2), W.rrdbnet_keys(23))
    ch, ref = _model(ctx, flat, 2, 23, CHAIN), _model(ctx, flat, 2, 23, NO_CHAIN | NO_RS)
    x = torch.from_numpy(smooth_u8(123, (1, 720, 1280, 3))).permute(0, 3, 1, 2).float().div(255.0).cuda()
    want = ref(x).clone()
    for i in range(12):
        got = ch(x)
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"run {i}: chain output changed"


def test_chain_two_contexts_on_two_streams_do_not_deadlock(ctx):
    """Two chains in flight at once (two contexts, two streams - the two-callers case): units come from a queue, so neither
    launch needs all of its workgroups resident; both finish and both are right."""
    flat = W.flatten(W.rrdbnet_table(3, scale=2, num_block=6), W.rrdbnet_keys(6))
    ctx2 = _capi.Context(0)
    m1, m2 = _model(ctx, flat, 2, 6, CHAIN), _model(ctx2, flat, 2, 6, CHAIN)
    ref = _model(ctx, flat, 2, 6, NO_CHAIN | NO_RS)
    x = torch.rand(1, 3, 720, 1280, generator=torch.Generator().manual_seed(8)).cuda()
    want = ref(x).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            outs.append((m1 if i % 2 == 0 else m2)(x))
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)
    del m2
    ctx2.close()


def test_service_model_flags_route_to_the_chain(ctx):
    """The drop-in service passes model_flags through to the library: a one-frame realesrgan job with
    model_flags=DEV_MODEL_CHAIN gives the uint8 frames of the default route within 1 LSB (conv5's summation order differs)."""
    from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
    tab = W.rrdbnet_table(13, scale=2)
    frames = torch.from_numpy(smooth_u8(7, (1, 72, 104, 3))).cuda()
    outs = []
    for fl in (0, CHAIN):
        svc = HipUpscalerService(device=0, denoising=False, upscaler_model="realesrgan", scale=2, lr_shape=(72, 104),
                                 model_name="RealESRGAN_x2plus", weights={"sr": tab}, lr_hr_resize=False, model_flags=fl)
        svc.proc_init()
        outs.append(svc.upscale(frames).cpu().to(torch.int16))
    assert outs[0].shape == (1, 144, 208, 3)
    assert int((outs[0] - outs[1]).abs().max()) <= 1


def test_chain_timeout_is_reported_once_and_never_lost(ctx, monkeypatch):
    """The chain's asynchronous failure mode (a unit gives up waiting) through the C ABI: injected in the dev library by letting
    units give up after ONE poll (SS4K_CHAIN_SPIN_LIMIT=1).  The sticky word lives in pinned host memory, no launch resets it:
    ss4k_model_check(wait=1) reports the failed forward before its output is used, a second check is clean, a failure that
    nobody checked for is reported by the next forward, and a healthy forward afterwards is bit-identical to a healthy model."""
    import ctypes as C
    from sharkshark4k_amd import build as B
    L = _capi.load(B.LIB_DEV)
    t = W.rrdbnet_table(5, scale=2, num_block=2)
    flat = np.ascontiguousarray(W.flatten(t, W.rrdbnet_keys(2)), dtype=np.float32)
    desc = _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2, flags=CHAIN)
    x = torch.rand(1, 3, 288, 416, generator=torch.Generator().manual_seed(3)).cuda()
    want = _capi.Model(ctx, desc, flat)(x).clone()
    hctx, hm = C.c_void_p(), C.c_void_p()
    assert L.ss4k_ctx_create(0, C.byref(hctx)) == 0
    assert L.ss4k_model_create(hctx, C.byref(desc), flat.ctypes.data_as(C.c_void_p), flat.size, C.byref(hm)) == 0, L.ss4k_last_error()
    out = torch.empty_like(want)
    st = int(torch.cuda.current_stream().cuda_stream)
    fwd = lambda: L.ss4k_model_forward(hm, x.data_ptr(), out.data_ptr(), 1, 288, 416, st)
    assert L.ss4k_model_check(hm, 1) == 0                      # nothing launched yet
    assert fwd() == 0 and L.ss4k_model_check(hm, 1) == 0       # healthy
    torch.cuda.synchronize(); assert torch.equal(out, want)
    monkeypatch.setenv("SS4K_CHAIN_SPIN_LIMIT", "1")           # read per chain launch (dev library)
    assert fwd() == 0                                          # the failure is asynchronous: enqueueing succeeds
    assert L.ss4k_model_check(hm, 1) != 0 and b"timed out" in L.ss4k_last_error()
    assert L.ss4k_model_check(hm, 1) == 0                      # reported once, then cleared by the code that reported it
    assert fwd() == 0                                          # a second failed forward that nobody checks ...
    torch.cuda.synchronize()
    monkeypatch.delenv("SS4K_CHAIN_SPIN_LIMIT")
    assert fwd() != 0 and b"timed out" in L.ss4k_last_error()  # ... is reported by the next forward, not lost
    assert fwd() == 0 and L.ss4k_model_check(hm, 1) == 0
    torch.cuda.synchronize(); assert torch.equal(out, want)
    L.ss4k_model_destroy(hm)
