"""``HipUpscalerService``: drop-in for the reference's ``FsrcnnUpscalerService``.

Same constructor arguments, attributes (``lr_shape``, ``output_shape``, ``on_queue``,
``exit_on_error``, ``device``, ``single_mode`` ...) and ``upscale(frames u8 NHWC) -> u8 NHWC``
contract as ``src/upscale/fsrcnn_upscaler.py:86-326``; all arithmetic runs in libss4k_hip.so
(``jit_mode='hip'``, the one backend of this build).  Extra keyword arguments that the reference
hard-codes are exposed: ``scale`` (FSRCNN factor, reference: always 4, fsrcnn_upscaler.py:101),
``model_name`` (reference: ``ArgsData.model_name``, realesrgan/factory.py:88), ``dtype`` and an
``lr_shape`` override.

Weights.  The reference always loads real checkpoints (``fsrcnn_x4-T91.pth``, the downloaded
RealESRGAN ``.pth`` files with the DNI blend, ``bsvd-32.pth``); so does this service:

* ``weights=None`` (default): the reference's own file names are looked up in ``checkpoint_dir``
  (else ``$SS4K_CHECKPOINT_DIR``); ``proc_init`` raises ``FileNotFoundError`` when one is missing.
* ``weights={'sr': X, 'sr_wdn': X, 'denoise': X}`` with X a ``.pth`` path, the dict ``torch.load``
  returns (``state_dict`` / ``params_ema`` / ``params``; BSVD's ``nets_list`` remap is applied) or a
  state-dict table; missing entries fall back to the ``checkpoint_dir`` lookup.
* ``weights='synthetic'``: deterministic generated weights for every model - an explicit opt-in
  used by the tests and ``bench.py`` only (logged loudly: the frames are noise).
"""
from __future__ import annotations

from typing import Mapping, Optional

import torch

from ..util.profiler import Profiler
from .upscaler_base import BaseUpscalerService, UpscalerQueueEntry  # noqa: F401

LR_LEVELS = [(360, 640), (540, 960), (630, 1120), (720, 1280), (900, 1600), (1080, 1920)]


def log(*args, **kwargs):
    print(f"HipUpscalerService: {' '.join(str(a) for a in args)}", **kwargs)


class HipUpscalerService(BaseUpscalerService):
    profiler: Profiler

    def __init__(self, lr_level=3, device=0, on_queue=None, denoising=True, denoise_rate=1.0,
                 upscaler_model="realesrgan", batch_size=1, jit_mode="hip", lr_hr_resize=True,
                 # knobs the reference hard-codes
                 scale=4, model_name=None, dtype="f16", weights=None, checkpoint_dir: Optional[str] = None,
                 lr_shape=None, single_mode=None, seed=0, model_flags=0, fsrcnn_dtype="f32"):
        if jit_mode not in (None, "hip"):
            raise Exception(f"jit_mode={jit_mode!r}: this build has one backend, 'hip'")
        if upscaler_model not in ("fsrcnn", "realesrgan"):
            raise Exception(upscaler_model)
        self.lr_shape = tuple(lr_shape) if lr_shape is not None else LR_LEVELS[lr_level]
        self.scale = scale
        self.denoise_rate = denoise_rate
        self.hr_shape = (1440, 2560)  # set and never read by the reference either (fsrcnn_upscaler.py:104)
        self.device = device
        self.on_queue = on_queue
        self.output_shape = None
        self.upscaler_model = upscaler_model
        self.single_mode = (upscaler_model != "realesrgan") if single_mode is None else bool(single_mode)
        self.denoising = denoising
        self.batch_size = batch_size
        self.jit_mode = "hip"
        self.lr_hr_resize = lr_hr_resize
        self.model_name = model_name
        self.dtype = dtype
        if not (weights is None or weights == "synthetic" or isinstance(weights, Mapping)):
            raise TypeError("weights must be None (checkpoint_dir lookup), 'synthetic' or {'sr'|'sr_wdn'|'denoise': path | checkpoint | table}")
        self.weights = weights if isinstance(weights, str) or weights is None else dict(weights)
        self.checkpoint_dir = checkpoint_dir
        self.seed = seed
        self.fsrcnn_dtype = fsrcnn_dtype  # 'f32': fp32-accurate (parity bar); 'f16': the reference engine's precision, ~2x the rate
        self.model_flags = int(model_flags)  # SS4K_MODEL_* routing switches for the SR model (include/ss4k.h)
        super().__init__()

    # worker side -----------------------------------------------------------------------------
    def proc_init(self):
        from .. import _capi
        from . import model as factory
        log("proc init")
        self.ctx = _capi.Context(self.device)
        self.torch_device = self.ctx.device
        if self.weights == "synthetic":
            log("WARNING: weights='synthetic' - every network runs on generated weights, output frames are noise")
        def spec(name):
            return "synthetic" if self.weights == "synthetic" else (self.weights or {}).get(name)
        if self.upscaler_model == "fsrcnn":
            self.model = factory.build_model_fsrcnn(self.ctx, factor=self.scale, weights=spec("sr"), seed=self.seed,
                                                    checkpoint_dir=self.checkpoint_dir, dtype=self.fsrcnn_dtype, flags=self.model_flags)
        else:
            self.model = factory.build_model_esrgan(
                self.ctx, model_name=self.model_name or factory.DEFAULT_REALESRGAN, denoise_rate=self.denoise_rate,
                weights=spec("sr"), weights_wdn=spec("sr_wdn") if self.weights != "synthetic" else None, dtype=self.dtype,
                seed=self.seed, checkpoint_dir=self.checkpoint_dir, flags=self.model_flags)
        self.denoise_model = None
        # quirk kept from the reference: with 'realesrgan' the batched path never denoises even when
        # denoising=True (fsrcnn_upscaler.py:109,168-233); the BSVD model is only used per-frame.
        if self.denoising and self.single_mode:
            self.denoise_model = factory.build_denoise_model(self.ctx, weights=spec("denoise"), dtype=self.dtype,
                                                             seed=self.seed, checkpoint_dir=self.checkpoint_dir)
        self._upscaler = None
        self._upscaler_key = None
        log("model loaded")

    def _get_upscaler(self):
        from .. import _capi
        key = (tuple(self.lr_shape), None if self.output_shape is None else tuple(self.output_shape),
               bool(self.lr_hr_resize), bool(self.single_mode), float(self.denoise_rate))
        if self._upscaler is None or key != self._upscaler_key:
            self._upscaler = _capi.Upscaler(self.ctx, self.model, self.lr_shape, self.output_shape, self.lr_hr_resize,
                                            self.single_mode, self.denoise_model, self.denoise_rate)
            self._upscaler_key = key
        return self._upscaler

    def proc_cleanup(self):
        pass

    def upscale(self, frames: torch.Tensor):
        assert isinstance(frames, torch.Tensor)
        if frames.device != self.torch_device:
            frames = frames.to(self.torch_device, non_blocking=True)
        if frames.ndim == 4:
            assert frames.shape[-1] == 3
            from .. import _capi
            prof = getattr(self, "profiler", None)
            up = self._get_upscaler()
            out = up(frames)
            if getattr(self, "model_flags", 0) & _capi.MODEL_CHAIN:
                # SS4K_MODEL_CHAIN's one asynchronous failure mode (a work unit timed out): the frames leave this worker right after
                # this call, so the status of THIS job's launch is awaited here (include/ss4k.h: ss4k_model_check, wait = 1)
                self.model.check(wait=True)
            if prof is not None:
                # the reference's span keys (fsrcnn_upscaler.py:276-278,290-300): host time around the
                # asynchronous stage launches, measured inside the library
                denoise_ms, model_ms = up.last_enqueue_ms()
                if self.denoise_model is not None:
                    prof.add("fsrcnn.denoise", denoise_ms / 1000.0)
                prof.add("fsrcnn.model", model_ms / 1000.0)
            return out
        raise Exception(frames.shape)
