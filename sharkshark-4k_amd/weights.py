"""Deterministic weight tables for the networks on the upscale path.

The reference downloads (RealESRGAN, ``src/upscale/model/realesrgan/factory.py:144-150``) or
lacks (BSVD, ``.MISSING_LARGE_BLOBS``) its checkpoints, so every box regenerates the same
synthetic weights from a counter-based generator (numpy Philox keyed by tensor name).  Tables
are plain ``{state_dict_key: float32 ndarray}`` in the reference's own key names and OIHW
layout, which is what ``ss4k_model_create`` (include/ss4k.h) takes as a flat blob.

Only numpy is used here: this module is data generation, not compute.
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import numpy as np

Table = "OrderedDict[str, np.ndarray]"


def _rng(seed: int, name: str) -> np.random.Generator:
    digest = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    key = int.from_bytes(digest[:8], "little")
    return np.random.Generator(np.random.Philox(key=key))


def _normal(seed: int, name: str, shape: Tuple[int, ...], std: float) -> np.ndarray:
    return (_rng(seed, name).standard_normal(shape) * std).astype(np.float32)


def _conv(tab, seed, name, cout, cin, k, gain=1.0, bias_std=0.01, kh=None):
    kh = k if kh is None else kh
    fan_in = cin * k * kh
    tab[name + ".weight"] = _normal(seed, name + ".weight", (cout, cin, kh, k),
                                    gain * math.sqrt(2.0 / fan_in))
    tab[name + ".bias"] = _normal(seed, name + ".bias", (cout,), bias_std)


def _prelu(tab, seed, name, ch):
    # nn.PReLU default slope is 0.25; jitter it so per-channel indexing is exercised.
    tab[name + ".weight"] = (0.25 + _normal(seed, name + ".weight", (ch,), 0.05)).astype(np.float32)


# --------------------------------------------------------------------------------------
# FSRCNN  (reference: src/upscale/model/fsrcnn/model.py:14-46)
# --------------------------------------------------------------------------------------
def fsrcnn_keys() -> List[str]:
    keys = ["feature_extraction.0.weight", "feature_extraction.0.bias", "feature_extraction.1.weight",
            "shrink.0.weight", "shrink.0.bias", "shrink.1.weight"]
    for i in range(4):
        keys += [f"map.{2*i}.weight", f"map.{2*i}.bias", f"map.{2*i+1}.weight"]
    keys += ["expand.0.weight", "expand.0.bias", "expand.1.weight", "deconv.weight", "deconv.bias"]
    return keys


def fsrcnn_table(seed: int = 0) -> Table:
    t: Table = OrderedDict()
    _conv(t, seed, "feature_extraction.0", 56, 1, 5)
    _prelu(t, seed, "feature_extraction.1", 56)
    _conv(t, seed, "shrink.0", 12, 56, 1)
    _prelu(t, seed, "shrink.1", 12)
    for i in range(4):
        _conv(t, seed, f"map.{2*i}", 12, 12, 3)
        _prelu(t, seed, f"map.{2*i+1}", 12)
    _conv(t, seed, "expand.0", 56, 12, 1)
    _prelu(t, seed, "expand.1", 56)
    # ConvTranspose2d weight is (C_in, C_out, kH, kW) = (56, 1, 9, 9)  (model.py:46)
    t["deconv.weight"] = _normal(seed, "deconv.weight", (56, 1, 9, 9), 0.02)
    t["deconv.bias"] = _normal(seed, "deconv.bias", (1,), 0.01)
    return OrderedDict((k, t[k]) for k in fsrcnn_keys())


# --------------------------------------------------------------------------------------
# RRDBNet  ([external] basicsr.archs.rrdbnet_arch, instantiated realesrgan/factory.py:113-125)
# --------------------------------------------------------------------------------------
def rrdbnet_keys(num_block: int) -> List[str]:
    keys = ["conv_first.weight", "conv_first.bias"]
    for b in range(num_block):
        for r in (1, 2, 3):
            for c in range(1, 6):
                keys += [f"body.{b}.rdb{r}.conv{c}.weight", f"body.{b}.rdb{r}.conv{c}.bias"]
    for n in ("conv_body", "conv_up1", "conv_up2", "conv_hr", "conv_last"):
        keys += [n + ".weight", n + ".bias"]
    return keys


def rrdbnet_table(seed: int = 0, scale: int = 2, num_feat: int = 64, num_block: int = 23,
                  num_grow_ch: int = 32, num_in_ch: int = 3, num_out_ch: int = 3) -> Table:
    t: Table = OrderedDict()
    cin0 = num_in_ch * (4 if scale == 2 else 16 if scale == 1 else 1)
    _conv(t, seed, "conv_first", num_feat, cin0, 3)
    for b in range(num_block):
        for r in (1, 2, 3):
            for c in range(1, 6):
                cin = num_feat + (c - 1) * num_grow_ch
                cout = num_grow_ch if c < 5 else num_feat
                # BasicSR default_init_weights(scale=0.1): kaiming-normal * 0.1
                _conv(t, seed, f"body.{b}.rdb{r}.conv{c}", cout, cin, 3, gain=0.1 * 3.0)
    _conv(t, seed, "conv_body", num_feat, num_feat, 3, gain=0.5)
    _conv(t, seed, "conv_up1", num_feat, num_feat, 3)
    _conv(t, seed, "conv_up2", num_feat, num_feat, 3)
    _conv(t, seed, "conv_hr", num_feat, num_feat, 3)
    _conv(t, seed, "conv_last", num_out_ch, num_feat, 3, gain=0.5)
    return OrderedDict((k, t[k]) for k in rrdbnet_keys(num_block))


# --------------------------------------------------------------------------------------
# SRVGGNetCompact  (reference: src/upscale/model/realesrgan/factory.py:18-82)
# --------------------------------------------------------------------------------------
def srvgg_keys(num_conv: int) -> List[str]:
    keys = []
    for i in range(num_conv + 1):
        keys += [f"body.{2*i}.weight", f"body.{2*i}.bias", f"body.{2*i+1}.weight"]
    keys += [f"body.{2*num_conv+2}.weight", f"body.{2*num_conv+2}.bias"]
    return keys


def srvgg_table(seed: int = 0, num_feat: int = 64, num_conv: int = 32, upscale: int = 4,
                num_in_ch: int = 3, num_out_ch: int = 3) -> Table:
    t: Table = OrderedDict()
    _conv(t, seed, "body.0", num_feat, num_in_ch, 3)
    _prelu(t, seed, "body.1", num_feat)
    for i in range(1, num_conv + 1):
        _conv(t, seed, f"body.{2*i}", num_feat, num_feat, 3, gain=0.9)
        _prelu(t, seed, f"body.{2*i+1}", num_feat)
    _conv(t, seed, f"body.{2*num_conv+2}", num_out_ch * upscale * upscale, num_feat, 3, gain=0.1)
    return OrderedDict((k, t[k]) for k in srvgg_keys(num_conv))


def dni_blend(table_a: Table, table_b: Table, alpha: float) -> Table:
    """Deep-network-interpolation blend, reference realesrgan/factory.py:152-157 ([external]
    RealESRGANer.dni: ``alpha*a + (1-alpha)*b`` per tensor)."""
    return OrderedDict((k, (alpha * table_a[k] + (1.0 - alpha) * table_b[k]).astype(np.float32))
                       for k in table_a)


# --------------------------------------------------------------------------------------
# BSVD as built by the service (reference: bsvd/factory.py:31-36, bsvd/model.py:353-475)
# --------------------------------------------------------------------------------------
def _bsvd_denblock_convs(chns, in_ch, out_ch, interm_ch) -> List[Tuple[str, int, int]]:
    c0, c1, c2 = chns
    return [
        ("inc.convblock.0", interm_ch, in_ch),
        ("inc.convblock.3", c0, interm_ch),
        ("downc0.convblock.0", c1, c0),
        ("downc0.memconv.c1.op.conv", c1, c1),
        ("downc0.memconv.c2.op.conv", c1, c1),
        ("downc1.convblock.0", c2, c1),
        ("downc1.memconv.c1.op.conv", c2, c2),
        ("downc1.memconv.c2.op.conv", c2, c2),
        ("upc2.memconv.c1.op.conv", c2, c2),
        ("upc2.memconv.c2.op.conv", c2, c2),
        ("upc2.convblock.0", c1 * 4, c2),
        ("upc1.memconv.c1.op.conv", c1, c1),
        ("upc1.memconv.c2.op.conv", c1, c1),
        ("upc1.convblock.0", c0 * 4, c1),
        ("outc.convblock.0", c0, c0),
        ("outc.convblock.3", out_ch, c0),
    ]


def bsvd_layers(chns=(32, 64, 128), mid_ch=32, in_ch=4, out_ch=3, interm_ch=30):
    layers = []
    for blk, (ci, co) in (("temp1", (in_ch, mid_ch)), ("temp2", (mid_ch, out_ch))):
        for name, cout, cin in _bsvd_denblock_convs(chns, ci, co, interm_ch):
            layers.append((f"{blk}.{name}", cout, cin))
    return layers


def bsvd_keys(**kw) -> List[str]:
    keys = []
    for name, _, _ in bsvd_layers(**kw):
        keys += [name + ".weight", name + ".bias"]
    return keys


def bsvd_table(seed: int = 0, **kw) -> Table:
    t: Table = OrderedDict()
    for name, cout, cin in bsvd_layers(**kw):
        # kaiming_normal_(nonlinearity='relu') as in bsvd/model.py:393-396; output convs damped
        gain = 0.25 if name.endswith("outc.convblock.3") else 1.0
        _conv(t, seed, name, cout, cin, 3, gain=gain)
    return t


# --------------------------------------------------------------------------------------
# flat blob <-> table
# --------------------------------------------------------------------------------------
def flatten(table: Table, keys: Iterable[str]) -> np.ndarray:
    """Concatenate tensors in ``keys`` order (the order include/ss4k.h documents per model)."""
    return np.concatenate([np.ascontiguousarray(table[k], dtype=np.float32).reshape(-1) for k in keys])


def num_params(table: Table) -> int:
    return int(sum(v.size for v in table.values()))
