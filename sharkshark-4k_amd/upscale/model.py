"""Model factories: the ``jit_mode='hip'`` backend behind the reference's ``build_model`` seams.

Reference seams replaced (each returns a callable ``model(x: NCHW float) -> NCHW float``):
  * ``build_model_fsrcnn``  <- ``src/upscale/model/fsrcnn/factory.py:5-71``
  * ``build_model_esrgan``  <- ``src/upscale/model/realesrgan/factory.py:108-234``
  * ``build_denoise_model`` <- ``src/upscale/model/bsvd/factory.py:21-83``
The reference downloads / loads ``.pth`` checkpoints; here weights are passed in as state-dict
shaped tables (``{key: ndarray}``, see ``weights.py``) or generated deterministically when absent.
"""
from __future__ import annotations

from typing import Mapping, Optional

import numpy as np

from .. import _capi
from .. import weights as W

# name -> (arch, kwargs) exactly the table in realesrgan/factory.py:112-138
REALESRGAN_ZOO = {
    "RealESRGAN_x4plus": ("rrdbnet", dict(scale=4, num_feat=64, num_block=23, num_grow_ch=32)),
    "RealESRNet_x4plus": ("rrdbnet", dict(scale=4, num_feat=64, num_block=23, num_grow_ch=32)),
    "RealESRGAN_x4plus_anime_6B": ("rrdbnet", dict(scale=4, num_feat=64, num_block=6, num_grow_ch=32)),
    "RealESRGAN_x2plus": ("rrdbnet", dict(scale=2, num_feat=64, num_block=23, num_grow_ch=32)),
    "realesr-animevideov3": ("srvgg", dict(num_feat=64, num_conv=16, upscale=4)),
    "realesr-general-x4v3": ("srvgg", dict(num_feat=64, num_conv=32, upscale=4)),
}
DEFAULT_REALESRGAN = "realesr-general-x4v3"  # hard-coded in the reference, realesrgan/factory.py:88


def _dtype(dtype) -> int:
    if dtype in (_capi.F16, "f16", "fp16", "half"):
        return _capi.F16
    if dtype in (_capi.F32, "f32", "fp32", "float"):
        return _capi.F32
    raise ValueError(f"unknown dtype {dtype!r}")


def build_model_fsrcnn(ctx: _capi.Context, factor: int = 4, weights: Optional[Mapping] = None, seed: int = 0):
    table = weights if weights is not None else W.fsrcnn_table(seed)
    desc = _capi.make_desc(_capi.FSRCNN, _capi.F32, scale=factor)
    return _capi.Model(ctx, desc, W.flatten(table, W.fsrcnn_keys()))


def build_model_esrgan(ctx: _capi.Context, model_name: str = DEFAULT_REALESRGAN, denoise_rate: float = 0.5,
                       weights: Optional[Mapping] = None, dtype="f16", seed: int = 0, **arch_overrides):
    if model_name not in REALESRGAN_ZOO:
        raise Exception(model_name)
    arch, kw = REALESRGAN_ZOO[model_name]
    kw = dict(kw, **arch_overrides)
    if arch == "rrdbnet":
        table = weights if weights is not None else W.rrdbnet_table(seed, **kw)
        desc = _capi.make_desc(_capi.RRDBNET, _dtype(dtype), scale=kw["scale"], num_feat=kw["num_feat"],
                               num_block=kw["num_block"], num_grow_ch=kw["num_grow_ch"])
        return _capi.Model(ctx, desc, W.flatten(table, W.rrdbnet_keys(kw["num_block"])))
    if weights is None:
        table = W.srvgg_table(seed, **kw)
        if model_name == "realesr-general-x4v3" and denoise_rate != 1:
            # DNI blend of the plain and the "wdn" checkpoints (realesrgan/factory.py:152-157)
            table = W.dni_blend(table, W.srvgg_table(seed + 1, **kw), denoise_rate)
    else:
        table = weights
    desc = _capi.make_desc(_capi.SRVGG, _dtype(dtype), scale=kw["upscale"], num_feat=kw["num_feat"],
                           num_block=kw["num_conv"])
    return _capi.Model(ctx, desc, W.flatten(table, W.srvgg_keys(kw["num_conv"])))


BSVD_VARIANTS = {  # bsvd/factory.py:31-36 (the one the service builds) and :94-98 / the commented-out block at :26-30
    "bsvd-32": dict(chns=(32, 64, 128), mid_ch=32, interm_ch=30),
    "bsvd-64": dict(chns=(64, 128, 256), mid_ch=64, interm_ch=64),
}


def build_denoise_model(ctx: _capi.Context, weights: Optional[Mapping] = None, dtype="f16", seed: int = 0,
                        stream: bool = False, variant: str = "bsvd-32"):
    """``stream=False``: the model the service calls, one independent frame per call (F = 1,
    ``fsrcnn_upscaler.py:277``).  ``stream=True``: ``BSVD.forward`` on ``(N,F,4,H,W)`` clips, all N*F
    frames run through the bidirectional buffers as one stream (``bsvd/model.py:515-580``)."""
    kw = BSVD_VARIANTS[variant]
    table = weights if weights is not None else W.bsvd_table(seed, **kw)
    desc = _capi.make_desc(_capi.BSVD, _dtype(dtype), scale=1, bsvd_stream=stream, bsvd_chns=kw["chns"],
                           bsvd_mid_ch=kw["mid_ch"], bsvd_interm_ch=kw["interm_ch"])
    return _capi.Model(ctx, desc, W.flatten(table, W.bsvd_keys(**kw)))
