"""Shared helpers for the parity tests (oracle = checker, never the thing under test)."""
import numpy as np
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import weights as W
from oracle import nets as onets

RTOL, ATOL = 1e-3, 1e-4  # north_star tolerance for the fp32 path


def assert_close(got, want, rtol=RTOL, atol=ATOL, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    tol = atol + rtol * np.abs(want.astype(np.float64))
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} outside rtol={rtol} atol={atol}; "
                           f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}")


def psnr(a, b, peak=1.0):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.from_numpy(np.asarray(b)).double()
    mse = torch.mean((a - b) ** 2).item()
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)


def assert_u8_close(got, want, max_lsb=1, max_frac=0.002, what=""):
    """uint8 frames after truncation: float parity within 1e-4 can flip the integer by one LSB, and only
    where the float sits within ~1e-4*255 of an integer: at most 0.2 % of the bytes (measured: 0.01-0.05 %)."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else np.asarray(want)
    assert got.shape == want.shape and got.dtype == np.uint8, f"{what}: {got.shape} {got.dtype} vs {want.shape}"
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= max_lsb, f"{what}: max |delta| = {d.max()} LSB"
    frac = float((d > 0).mean())
    assert frac <= max_frac, f"{what}: {frac:.4f} of the bytes differ"


def rrdb_small_table(seed=5, scale=2, num_block=2):
    return W.rrdbnet_table(seed, scale=scale, num_feat=64, num_block=num_block, num_grow_ch=32)


def srvgg_full_table(seed, seed_b=None, alpha=0.3):
    """SRVGGNetCompact at the depth the reference ships (num_feat 64, num_conv 32, x4: realesrgan/factory.py:88,132-138).  With seed_b: the
    DNI blend of two generated checkpoints (factory.py:152-157) whose PReLU slopes are redrawn from [-0.5, 1.7] on every other layer and
    from [-0.5, 1.0] on the rest, so that both forms of the HIP epilogue (max(t, t s) needs every slope <= 1) run at depth."""
    t = W.srvgg_table(seed, num_feat=64, num_conv=32, upscale=4)
    if seed_b is None:
        return t
    t = W.dni_blend(t, W.srvgg_table(seed_b, num_feat=64, num_conv=32, upscale=4), alpha)
    rng = np.random.default_rng(1000 + seed)
    i = 0
    for k in list(t):
        if np.asarray(t[k]).ndim == 1 and k.endswith(".weight"):   # PReLU slopes
            lo, hi = (-0.5, 1.7) if i % 2 == 0 else (-0.5, 1.0)
            # mean slope 0.6 / 0.25 keeps the 33-layer chain from blowing up or dying out
            t[k] = rng.uniform(lo, hi, t[k].shape).astype(np.float32)
            i += 1
    return t


def srvgg_table_for(m):
    """The SRVGG weight table a golden vector was generated with, from its MANIFEST entry (tests/golden/make_golden.py)."""
    wspec = m.get("weights", "")
    if "srvgg_full_table" in wspec:
        a, b = wspec.split("srvgg_full_table(")[1].rstrip(")").split(",")
        return srvgg_full_table(int(a), None if b.strip() == "None" else int(b))
    seed = m["seed"] if "seed" in m else int(wspec.split("seed=")[1].rstrip(")"))
    return W.srvgg_table(seed=seed, num_feat=m["num_feat"], num_conv=m["num_conv"], upscale=m["upscale"])


def smooth_u8(seed, shape):
    n, h, w, c = shape
    g = np.random.default_rng(seed).random((n, h + 8, w + 8, c)).astype(np.float32)
    k = 9
    cs = np.cumsum(np.cumsum(np.pad(g, ((0, 0), (1, 0), (1, 0), (0, 0))), 1), 2)
    box = (cs[:, k:, k:] - cs[:, :-k, k:] - cs[:, k:, :-k] + cs[:, :-k, :-k]) / (k * k)
    box = (box - box.min()) / (box.max() - box.min())
    return (box[:, :h, :w] * 255).astype(np.uint8)


def record_measured(name, **values):
    """Parity figures a test measured (PSNR, max LSB, ...) are appended to gpurun_out/parity_measured.json, so that the
    asserted thresholds can be read next to what was measured (tools/parity_measured.sh copies the file to profiles/)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "gpurun_out", "parity_measured.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.exists(path):
            with open(path) as f:
                data = json.load(f)
        data[name] = {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in values.items()}
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
