"""Model factories: the ``jit_mode='hip'`` backend behind the reference's ``build_model`` seams.

Reference seams replaced (each returns a callable ``model(x: NCHW float) -> NCHW float``):
  * ``build_model_fsrcnn``  <- ``src/upscale/model/fsrcnn/factory.py:5-71``
  * ``build_model_esrgan``  <- ``src/upscale/model/realesrgan/factory.py:108-234``
  * ``build_denoise_model`` <- ``src/upscale/model/bsvd/factory.py:21-83``
The reference downloads / loads ``.pth`` checkpoints (``fsrcnn/factory.py:8-13``,
``realesrgan/factory.py:140-170``, ``bsvd/factory.py:31-36``).  Here every factory takes ``weights``:

* a path to such a ``.pth`` file, or the dict ``torch.load`` returns for it (``state_dict`` /
  ``params_ema`` / ``params``; BSVD's ``nets_list`` prefixes and ``convblock``->``memconv`` remap are
  applied) - routed through ``checkpoints.*_from_checkpoint``;
* a state-dict shaped table ``{key: ndarray}`` in the reference's key names (``weights.py``);
* ``None``: the reference's own file names are looked up in ``checkpoint_dir`` (argument, else
  ``$SS4K_CHECKPOINT_DIR``); a missing file raises - a service never runs on made-up weights silently;
* the string ``'synthetic'``: deterministic generated weights.  An explicit opt-in for tests and
  ``bench.py`` (there is no network on the GPU boxes to fetch checkpoints from).
"""
from __future__ import annotations

import os
from typing import Mapping, Optional, Union

import numpy as np

from .. import _capi
from .. import checkpoints as CK
from .. import weights as W

SYNTHETIC = "synthetic"
WeightSpec = Union[None, str, "os.PathLike", Mapping]

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


def _load_checkpoint(spec, what: str, default_name: str, checkpoint_dir: Optional[str]):
    """-> (kind, object): kind 'table' (already in the reference's key names) or 'ckpt' (torch.load result)."""
    if spec is None:
        d = checkpoint_dir or os.environ.get("SS4K_CHECKPOINT_DIR")
        path = os.path.join(d, default_name) if d else None
        if not path or not os.path.isfile(path):
            raise FileNotFoundError(
                f"{what}: no weights given and {path or default_name!r} not found. Pass weights=<path to the .pth | "
                f"torch.load(...) dict | state-dict table>, set checkpoint_dir / $SS4K_CHECKPOINT_DIR to the directory "
                f"holding {default_name}, or pass weights='synthetic' to run on generated weights (tests / bench only)")
        spec = path
    if isinstance(spec, (str, os.PathLike)):
        import torch
        # tensors-only unpickling: the reference's checkpoints (FSRCNN T91, RealESRGAN, BSVD) are plain tensor dicts, and a
        # .pth found in a directory is not trusted code.  SS4K_UNSAFE_TORCH_LOAD=1 is the explicit opt-in for a legacy file
        # that pickles other objects (the reference itself loads with torch's old default, fsrcnn/factory.py:12).
        unsafe = os.environ.get("SS4K_UNSAFE_TORCH_LOAD") == "1"
        try:
            return "ckpt", torch.load(os.fspath(spec), map_location="cpu", weights_only=not unsafe)
        except Exception as e:  # pickle.UnpicklingError (torch re-raises it with its own advice to pass weights_only=False)
            if unsafe or "weights_only" not in str(e).lower() and "unpickl" not in type(e).__name__.lower():
                raise
            raise RuntimeError(
                f"{what}: {os.fspath(spec)!r} pickles objects other than tensors and plain containers, and checkpoints are "
                f"loaded tensors-only. If you trust the file (it can run code when unpickled), set SS4K_UNSAFE_TORCH_LOAD=1 "
                f"- the reference loads with that permissive default (fsrcnn/factory.py:8-13, bsvd/model.py:488). ({e})") from e
    if isinstance(spec, Mapping):
        if any(k in spec for k in ("state_dict", "params_ema", "params")) or any("nets_list." in str(k) for k in spec):
            return "ckpt", spec
        return "table", spec
    raise TypeError(f"{what}: weights must be a path, a checkpoint dict, a state-dict table, None or 'synthetic'; got {type(spec)}")


def fsrcnn_table_from(weights: WeightSpec, factor: int = 4, seed: int = 0, checkpoint_dir: Optional[str] = None):
    if isinstance(weights, str) and weights == SYNTHETIC:
        return W.fsrcnn_table(seed)
    kind, obj = _load_checkpoint(weights, "FSRCNN", f"fsrcnn_x{factor}-T91.pth", checkpoint_dir)  # fsrcnn/factory.py:8-10
    return CK.fsrcnn_from_checkpoint(obj) if kind == "ckpt" else obj


# Every factory is two halves: ``*_desc`` (the model description: host-only, needs no checkpoint - every rank of a node can size the
# weight blob from it) and ``*_flat`` (the flat fp32 state_dict blob: reads / blends / generates the weights - on a multi-GPU node only
# rank 0 does that and the blob reaches the others through ``sharding.broadcast_weights``, see ``HipUpscalerService.proc_init``).
def fsrcnn_desc(factor: int = 4, dtype="f32", flags: int = 0):
    return _capi.make_desc(_capi.FSRCNN, _dtype(dtype), scale=factor, flags=flags)


def fsrcnn_flat(factor: int = 4, weights: WeightSpec = None, seed: int = 0, checkpoint_dir: Optional[str] = None) -> np.ndarray:
    return W.flatten(fsrcnn_table_from(weights, factor, seed, checkpoint_dir), W.fsrcnn_keys())


def build_model_fsrcnn(ctx: _capi.Context, factor: int = 4, weights: WeightSpec = None, seed: int = 0,
                       checkpoint_dir: Optional[str] = None, dtype="f32", flags: int = 0):
    """dtype 'f32' (default): fp32 accuracy (the 1e-3 / 1e-4 parity bar against the CPU forward).  'f16': fp16 operands with
    fp32 accumulation - the precision the reference's TensorRT engine runs this network in (fsrcnn/factory.py:47-69)."""
    return _capi.Model(ctx, fsrcnn_desc(factor, dtype, flags), fsrcnn_flat(factor, weights, seed, checkpoint_dir))


def esrgan_table_from(model_name: str = DEFAULT_REALESRGAN, denoise_rate: float = 0.5, weights: WeightSpec = None,
                      seed: int = 0, weights_wdn: WeightSpec = None, checkpoint_dir: Optional[str] = None, **arch_overrides):
    """-> (arch, arch kwargs, state-dict table).  ``weights_wdn``: the second ('wdn') checkpoint of
    ``realesr-general-x4v3``; with ``denoise_rate != 1`` the network is the DNI blend
    ``denoise_rate * general + (1 - denoise_rate) * wdn`` (realesrgan/factory.py:152-157).  A ready-made
    table passed as ``weights`` without a wdn partner is used as it is."""
    if model_name not in REALESRGAN_ZOO:
        raise Exception(model_name)
    arch, kw = REALESRGAN_ZOO[model_name]
    kw = dict(kw, **arch_overrides)
    synthetic = isinstance(weights, str) and weights == SYNTHETIC
    if arch == "rrdbnet":
        if synthetic:
            return arch, kw, W.rrdbnet_table(seed, **kw)
        kind, obj = _load_checkpoint(weights, model_name, model_name + ".pth", checkpoint_dir)
        return arch, kw, (CK.realesrgan_from_checkpoint(obj, "rrdbnet", num_block=kw["num_block"]) if kind == "ckpt" else obj)
    blend = model_name == "realesr-general-x4v3" and denoise_rate != 1
    if synthetic:
        table = W.srvgg_table(seed, **kw)
        if blend:
            table = W.dni_blend(table, W.srvgg_table(seed + 1, **kw), denoise_rate)
        return arch, kw, table
    kind, obj = _load_checkpoint(weights, model_name, model_name + ".pth", checkpoint_dir)
    table = CK.realesrgan_from_checkpoint(obj, "srvgg", num_conv=kw["num_conv"]) if kind == "ckpt" else obj
    if blend and (kind == "ckpt" or weights_wdn is not None):
        # a checkpoint is one of the two DNI operands: its partner is required, as in the reference
        k2, o2 = _load_checkpoint(weights_wdn, model_name + " (wdn)", "realesr-general-wdn-x4v3.pth", checkpoint_dir)
        wdn = CK.realesrgan_from_checkpoint(o2, "srvgg", num_conv=kw["num_conv"]) if k2 == "ckpt" else o2
        table = W.dni_blend(table, wdn, denoise_rate)
    return arch, kw, table


def esrgan_desc(model_name: str = DEFAULT_REALESRGAN, dtype="f16", flags: int = 0, **arch_overrides):
    if model_name not in REALESRGAN_ZOO:
        raise Exception(model_name)
    arch, kw = REALESRGAN_ZOO[model_name]
    kw = dict(kw, **arch_overrides)
    if arch == "rrdbnet":
        return _capi.make_desc(_capi.RRDBNET, _dtype(dtype), scale=kw["scale"], num_feat=kw["num_feat"],
                               num_block=kw["num_block"], num_grow_ch=kw["num_grow_ch"], flags=flags)
    return _capi.make_desc(_capi.SRVGG, _dtype(dtype), scale=kw["upscale"], num_feat=kw["num_feat"],
                           num_block=kw["num_conv"], flags=flags)


def esrgan_flat(model_name: str = DEFAULT_REALESRGAN, denoise_rate: float = 0.5, weights: WeightSpec = None, seed: int = 0,
                weights_wdn: WeightSpec = None, checkpoint_dir: Optional[str] = None, **arch_overrides) -> np.ndarray:
    arch, kw, table = esrgan_table_from(model_name, denoise_rate, weights, seed, weights_wdn, checkpoint_dir, **arch_overrides)
    return W.flatten(table, W.rrdbnet_keys(kw["num_block"]) if arch == "rrdbnet" else W.srvgg_keys(kw["num_conv"]))


def build_model_esrgan(ctx: _capi.Context, model_name: str = DEFAULT_REALESRGAN, denoise_rate: float = 0.5,
                       weights: WeightSpec = None, dtype="f16", seed: int = 0, weights_wdn: WeightSpec = None,
                       checkpoint_dir: Optional[str] = None, flags: int = 0, **arch_overrides):
    """flags: SS4K_MODEL_* routing switches (include/ss4k.h)."""
    return _capi.Model(ctx, esrgan_desc(model_name, dtype, flags, **arch_overrides),
                       esrgan_flat(model_name, denoise_rate, weights, seed, weights_wdn, checkpoint_dir, **arch_overrides))


BSVD_VARIANTS = {  # bsvd/factory.py:31-36 (the one the service builds) and :94-98 / the commented-out block at :26-30
    "bsvd-32": dict(chns=(32, 64, 128), mid_ch=32, interm_ch=30),
    "bsvd-64": dict(chns=(64, 128, 256), mid_ch=64, interm_ch=64),
}


def bsvd_table_from(weights: WeightSpec, seed: int = 0, variant: str = "bsvd-32", checkpoint_dir: Optional[str] = None):
    kw = BSVD_VARIANTS[variant]
    if isinstance(weights, str) and weights == SYNTHETIC:
        return W.bsvd_table(seed, **kw)
    kind, obj = _load_checkpoint(weights, "BSVD", variant + ".pth", checkpoint_dir)  # bsvd/factory.py:35
    return CK.bsvd_from_checkpoint(obj, **kw) if kind == "ckpt" else obj


def denoise_desc(dtype="f16", stream: bool = False, variant: str = "bsvd-32", flags: int = 0):
    kw = BSVD_VARIANTS[variant]
    return _capi.make_desc(_capi.BSVD, _dtype(dtype), scale=1, bsvd_stream=stream, bsvd_chns=kw["chns"],
                           bsvd_mid_ch=kw["mid_ch"], bsvd_interm_ch=kw["interm_ch"], flags=flags)


def denoise_flat(weights: WeightSpec = None, seed: int = 0, variant: str = "bsvd-32", checkpoint_dir: Optional[str] = None) -> np.ndarray:
    return W.flatten(bsvd_table_from(weights, seed, variant, checkpoint_dir), W.bsvd_keys(**BSVD_VARIANTS[variant]))


def build_denoise_model(ctx: _capi.Context, weights: WeightSpec = None, dtype="f16", seed: int = 0,
                        stream: bool = False, variant: str = "bsvd-32", checkpoint_dir: Optional[str] = None, flags: int = 0):
    """``stream=False``: the model the service calls, one independent frame per call (F = 1,
    ``fsrcnn_upscaler.py:277``).  ``stream=True``: ``BSVD.forward`` on ``(N,F,4,H,W)`` clips, all N*F
    frames run through the bidirectional buffers as one stream (``bsvd/model.py:515-580``)."""
    return _capi.Model(ctx, denoise_desc(dtype, stream, variant, flags), denoise_flat(weights, seed, variant, checkpoint_dir))
