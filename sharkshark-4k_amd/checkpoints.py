"""Checkpoint import: upstream ``.pth`` layouts -> the flat state_dict tables libss4k_hip takes
(SURVEY §8 f4).  Pure key/shape bookkeeping on the host; tensors may be torch tensors or ndarrays.

* FSRCNN: ``torch.load(path)['state_dict']`` (reference ``fsrcnn/factory.py:8-13``).
* RealESRGAN family: ``['params_ema']`` if present else ``['params']`` ([external] RealESRGANer
  behaviour the reference relies on, ``realesrgan/factory.py:160-170``); DNI blend via
  ``weights.dni_blend`` (``:152-157``).
* BSVD: ``['params']`` with prefixes ``[module.]base_model.nets_list.{0,1}.`` and the per-block
  remaps the reference applies in ``load_from`` (``bsvd/model.py:487-499``): ``DownBlock`` keeps
  ``convblock.0`` and turns ``convblock.3.{c1,c2}.net.`` into ``memconv.{c1,c2}.op.conv.``
  (``:276-279,167-169``); ``UpBlock`` turns ``convblock.0.{c1,c2}.net.`` into ``memconv...`` and
  ``convblock.1`` into ``convblock.0`` (``:304-306``).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Mapping

import numpy as np

from . import weights as W


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


def _ordered(src: Mapping, keys, what: str):
    missing = [k for k in keys if k not in src]
    if missing:
        raise KeyError(f"{what}: checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    return OrderedDict((k, _np(src[k])) for k in keys)


def fsrcnn_from_checkpoint(ckpt: Mapping):
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return _ordered(sd, W.fsrcnn_keys(), "FSRCNN")


def realesrgan_from_checkpoint(ckpt: Mapping, arch: str, **kw):
    sd = ckpt.get("params_ema", ckpt.get("params", ckpt))
    if arch == "rrdbnet":
        return _ordered(sd, W.rrdbnet_keys(kw.get("num_block", 23)), "RRDBNet")
    if arch == "srvgg":
        return _ordered(sd, W.srvgg_keys(kw.get("num_conv", 32)), "SRVGGNetCompact")
    raise ValueError(arch)


def _bsvd_block_remap(key: str) -> str:
    """Upstream DenBlock key -> reference module key (the names weights.bsvd_keys() uses)."""
    for blk in ("downc0.", "downc1."):
        if key.startswith(blk):
            rest = key[len(blk):]
            if rest.startswith("convblock.3."):
                return blk + "memconv." + rest[len("convblock.3."):].replace("net.", "op.conv.")
            return key
    for blk in ("upc2.", "upc1."):
        if key.startswith(blk):
            rest = key[len(blk):]
            if rest.startswith("convblock.0."):
                return blk + "memconv." + rest[len("convblock.0."):].replace("net.", "op.conv.")
            if rest.startswith("convblock.1."):
                return blk + "convblock.0." + rest[len("convblock.1."):]
            return key
    return key


def bsvd_from_checkpoint(ckpt: Mapping, **kw):
    sd = ckpt["params"] if "params" in ckpt else ckpt
    first = next(iter(sd))
    base = "module.base_model." if "module" in first else "base_model."
    out = {}
    for i, blk in enumerate(("temp1", "temp2")):
        prefix = f"{base}nets_list.{i}."
        for k, v in sd.items():
            if prefix in k:
                out[f"{blk}." + _bsvd_block_remap(k.replace(prefix, ""))] = v
    return _ordered(out, W.bsvd_keys(**kw), "BSVD")
