"""ctypes binding of libss4k_hip.so (C ABI declared in include/ss4k.h).

PyTorch is used only as the owner of device memory and streams: every function here takes
``torch`` CUDA(HIP) tensors, passes their ``data_ptr()`` and the current stream to the library and
returns tensors.  There is no CPU fallback: a missing library or a failing call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS4K_LIB") or os.path.join(_HERE, "libss4k_hip.so")  # override = A/B builds

FSRCNN, RRDBNET, SRVGG, BSVD = 1, 2, 3, 4
F32, F16 = 0, 1

# every symbol include/ss4k.h declares (tests check the library exports all of them)
SYMBOLS = [
    "ss4k_abi_version", "ss4k_last_error", "ss4k_ctx_create", "ss4k_ctx_destroy", "ss4k_ctx_device",
    "ss4k_model_param_count", "ss4k_model_create", "ss4k_model_destroy", "ss4k_model_out_shape",
    "ss4k_model_in_channels", "ss4k_model_forward", "ss4k_model_check", "ss4k_upscaler_create", "ss4k_upscaler_destroy",
    "ss4k_upscaler_reset", "ss4k_upscaler_out_shape", "ss4k_upscaler_last_enqueue_ms", "ss4k_model_workspace_bytes", "ss4k_upscale_frames", "ss4k_upscaler_enable_taps",
    "ss4k_upscaler_read_tap", "ss4k_op_u8nhwc_to_f32nchw", "ss4k_op_area_resize", "ss4k_op_bicubic_resize",
    "ss4k_op_bilinear_resize", "ss4k_op_depthwise_reflect", "ss4k_op_plane_stats", "ss4k_op_f32nchw_to_u8nhwc",
    "ss4k_prof_enable", "ss4k_prof_reset", "ss4k_prof_read", "ss4k_prof_read_kind", "ss4k_prof_read_family", "ss4k_prof_read_section_ms",
    "ss4k_stream_pair_check", "ss4k_op_cv_area_shape", "ss4k_op_cv_area_resize_u8",
]
DEV_SYMBOLS = ["ss4k_bench_conv"]  # include/ss4k_dev.h: libss4k_hip_dev.so only (SS4K_LIB=.../libss4k_hip_dev.so)


class ModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("scale", C.c_int32), ("num_feat", C.c_int32),
                ("num_block", C.c_int32), ("num_grow_ch", C.c_int32), ("bsvd_chns", C.c_int32 * 3),
                ("bsvd_mid_ch", C.c_int32), ("bsvd_interm_ch", C.c_int32), ("bsvd_stream", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32 * 3)]


# ss4k_model_desc.flags (include/ss4k.h)
MODEL_FS_EXACT, MODEL_ONE_CHAIN, MODEL_TWO_CHAINS, MODEL_TILE_ROWS_16, MODEL_TILE_ROWS_20 = 1, 2, 4, 16, 32
MODEL_NO_PAIR, MODEL_HR_F32, MODEL_NO_DENSE, MODEL_NO_WIDE, MODEL_NO_UPS_PRESUM, MODEL_NO_W16 = 256, 512, 1024, 4096, 8192, 32768
MODEL_FLAGS_ALL = 1 | 2 | 4 | 16 | 32 | 256 | 512 | 1024 | 4096 | 8192 | 32768
# include/ss4k_dev.h: accepted by libss4k_hip_dev.so only (kernels of rounds 2-4 that are on no product route: tools/dev_tests/)
DEV_MODEL_CHAIN, DEV_MODEL_CONV5_RS = 128, 16384


class UpscaleCfg(C.Structure):
    _fields_ = [("lr_h", C.c_int32), ("lr_w", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32),
                ("lr_hr_resize", C.c_int32), ("single_mode", C.c_int32), ("sr_is_realesrgan", C.c_int32),
                ("denoising", C.c_int32), ("reserved0", C.c_int32), ("denoise_rate", C.c_double),
                ("reserved", C.c_int32 * 6)]


class Ss4kError(RuntimeError):
    pass


_lib = None
#: set once this process has created a library context = initialised the HIP runtime (BaseService.start_method reads it: a child forked
#: from such a process cannot use the GPU)
GPU_TOUCHED = False


def lib() -> C.CDLL:
    """Load the HIP library; fail loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        _lib = load(LIB_PATH)
    return _lib


def load(path: str) -> C.CDLL:
    """Bind one build of the library (the product library, or libss4k_hip_dev.so for a test that needs its hooks)."""
    if not os.path.exists(path):
        raise Ss4kError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(path)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    L.ss4k_last_error.restype = C.c_char_p
    L.ss4k_ctx_create.argtypes = [i, C.POINTER(vp)]
    L.ss4k_ctx_destroy.argtypes = [vp]; L.ss4k_ctx_destroy.restype = None
    L.ss4k_ctx_device.argtypes = [vp]
    L.ss4k_model_param_count.argtypes = [C.POINTER(ModelDesc)]; L.ss4k_model_param_count.restype = sz
    L.ss4k_model_create.argtypes = [vp, C.POINTER(ModelDesc), vp, sz, C.POINTER(vp)]
    L.ss4k_model_destroy.argtypes = [vp]; L.ss4k_model_destroy.restype = None
    L.ss4k_model_out_shape.argtypes = [vp, i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.ss4k_model_in_channels.argtypes = [vp]
    L.ss4k_model_forward.argtypes = [vp, vp, vp, i, i, i, vp]
    if hasattr(L, "ss4k_model_check"):   # (absent from ABI-1 builds, which tools/lib_ab.py loads for A/B timing)
        L.ss4k_model_check.argtypes = [vp, i]
    L.ss4k_upscaler_create.argtypes = [vp, C.POINTER(UpscaleCfg), vp, vp, C.POINTER(vp)]
    L.ss4k_upscaler_destroy.argtypes = [vp]; L.ss4k_upscaler_destroy.restype = None
    L.ss4k_upscaler_reset.argtypes = [vp]
    L.ss4k_model_workspace_bytes.argtypes = [vp, i, i, i, C.POINTER(C.c_size_t)]
    L.ss4k_upscaler_last_enqueue_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ss4k_upscaler_out_shape.argtypes = [vp, i, i, i, C.POINTER(i), C.POINTER(i)]
    L.ss4k_upscale_frames.argtypes = [vp, vp, i, i, i, vp, sz, vp]
    L.ss4k_upscaler_enable_taps.argtypes = [vp, i]
    L.ss4k_upscaler_read_tap.argtypes = [vp, i, vp, sz, C.POINTER(i * 4), vp]
    L.ss4k_op_u8nhwc_to_f32nchw.argtypes = [vp, vp, vp, i, i, i, i, vp]
    for name in ("ss4k_op_area_resize", "ss4k_op_bicubic_resize", "ss4k_op_bilinear_resize"):
        getattr(L, name).argtypes = [vp, vp, vp, i, i, i, i, i, vp]
    L.ss4k_op_depthwise_reflect.argtypes = [vp, vp, vp, i, i, i, vp, i, vp]
    L.ss4k_op_plane_stats.argtypes = [vp, vp, vp, i, i, vp]
    L.ss4k_op_f32nchw_to_u8nhwc.argtypes = [vp, vp, vp, i, i, i, i, vp]
    if hasattr(L, "ss4k_bench_conv"):  # dev library only
        L.ss4k_bench_conv.argtypes = [vp, i, i, i, i, i, i, i, i, i, C.POINTER(C.c_double), vp]
    L.ss4k_prof_enable.argtypes = [vp, i]
    L.ss4k_prof_reset.argtypes = [vp]
    L.ss4k_prof_read.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "ss4k_prof_read_kind"):
        L.ss4k_prof_read_kind.argtypes = [vp, i, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "ss4k_prof_read_family"):
        L.ss4k_prof_read_family.argtypes = [vp, i, C.c_char_p, sz, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ss4k_prof_read_section_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.ss4k_stream_pair_check.argtypes = [vp, vp, vp, C.POINTER(C.c_int)]
    L.ss4k_op_cv_area_shape.argtypes = [i, i, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.ss4k_op_cv_area_resize_u8.argtypes = [vp, vp, vp, sz, i, i, i, i, C.c_double, C.c_double, vp]
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise Ss4kError(f"libss4k_hip error {rc}: {lib().ss4k_last_error().decode()}")


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


def _dev_index(device) -> int:
    d = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
    return d.index if d.index is not None else torch.cuda.current_device()


class Context:
    """One per GPU (include/ss4k.h: ss4k_ctx)."""

    def __init__(self, device=0):
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        global GPU_TOUCHED
        rc = lib().ss4k_ctx_create(self.device_index, C.byref(h))
        # (a box without any GPU has no runtime to initialise: the failed attempt leaves the process as fork-safe as it was)
        GPU_TOUCHED = GPU_TOUCHED or rc == 0 or torch.cuda.device_count() > 0
        _check(rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().ss4k_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- profiling hooks (bench.py roofline leg)
    def prof_enable(self, on: bool):
        _check(lib().ss4k_prof_enable(self._h, int(on)))

    def prof_reset(self):
        _check(lib().ss4k_prof_reset(self._h))

    def prof_read(self) -> Tuple[int, float, float]:
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _check(lib().ss4k_prof_read(self._h, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def prof_read_kind(self, kind: int) -> Tuple[int, float, float]:
        """(launches, total ms, algorithmic FLOPs) of one kernel family: 0 conv, 1 / 2 / 3 FSRCNN head / mapping / tail."""
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        _check(lib().ss4k_prof_read_kind(self._h, kind, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def prof_read_families(self):
        """[(kernel build, launches, total ms, algorithmic FLOPs)] of the conv launches since the last reset."""
        out, idx = [], 0
        while True:
            name, n, ms, fl = C.create_string_buffer(256), C.c_int64(), C.c_double(), C.c_double()
            if lib().ss4k_prof_read_family(self._h, idx, name, 256, C.byref(n), C.byref(ms), C.byref(fl)) != 0:
                return out
            out.append((name.value.decode(), n.value, ms.value, fl.value))
            idx += 1

    def streams_side_by_side(self, a, b) -> bool:
        """Measured (~ 3 ms, synchronises both): do the torch streams `a` and `b` run beside each other at full launch rate?"""
        ok = C.c_int()
        _check(lib().ss4k_stream_pair_check(self._h, int(a.cuda_stream), int(b.cuda_stream), C.byref(ok)))
        return bool(ok.value)

    def prof_read_section_ms(self) -> float:
        """Wall time of the profiled forwards' conv sections (first conv launch to the end of the last, caller's stream)."""
        ms = C.c_double()
        _check(lib().ss4k_prof_read_section_ms(self._h, C.byref(ms)))
        return ms.value

    def bench_conv(self, dtype, cin0, cin1, cout, n, h, w, flags=0, iters=20) -> float:
        if not hasattr(lib(), "ss4k_bench_conv"):
            raise Ss4kError("ss4k_bench_conv lives in libss4k_hip_dev.so: run with SS4K_LIB=<package>/libss4k_hip_dev.so")
        us = C.c_double()
        _check(lib().ss4k_bench_conv(self._h, dtype, cin0, cin1, cout, n, h, w, flags, iters, C.byref(us), _stream()))
        return us.value

    # -- granular ops (tests)
    def _f32(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return t

    def u8nhwc_to_f32nchw(self, x: torch.Tensor) -> torch.Tensor:
        assert x.is_cuda and x.dtype == torch.uint8 and x.is_contiguous() and x.ndim == 4
        n, h, w, c = x.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        _check(lib().ss4k_op_u8nhwc_to_f32nchw(self._h, x.data_ptr(), out.data_ptr(), n, h, w, c, _stream()))
        return out

    def _resize(self, fn, x, size):
        self._f32(x)
        n, c, h, w = x.shape
        out = torch.empty((n, c, size[0], size[1]), dtype=torch.float32, device=x.device)
        _check(fn(self._h, x.data_ptr(), out.data_ptr(), n * c, h, w, size[0], size[1], _stream()))
        return out

    def area_resize(self, x, size):
        return self._resize(lib().ss4k_op_area_resize, x, size)

    def bicubic_resize(self, x, size):
        return self._resize(lib().ss4k_op_bicubic_resize, x, size)

    def cv_area_resize(self, frames: torch.Tensor, fx: float, fy: Optional[float] = None) -> torch.Tensor:
        """``cv2.resize(frame, None, fx=fx, fy=fy, interpolation=cv2.INTER_AREA)`` of every frame of a uint8 NHWC device tensor (shrinking,
        non-integer 1 / f: the image server's pre / post scale, image_pipeline.py:272-273,347-348) -> uint8 NHWC."""
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.ndim == 4
        fy = fx if fy is None else fy
        n, h, w, c = frames.shape
        oh, ow = C.c_int(), C.c_int()
        _check(lib().ss4k_op_cv_area_shape(h, w, fx, fy, C.byref(oh), C.byref(ow)))
        out = torch.empty((n, oh.value, ow.value, c), dtype=torch.uint8, device=frames.device)
        _check(lib().ss4k_op_cv_area_resize_u8(self._h, frames.data_ptr(), out.data_ptr(), out.numel(), n, h, w, c, fx, fy, _stream()))
        return out

    def bilinear_resize(self, x, size):
        return self._resize(lib().ss4k_op_bilinear_resize, x, size)

    def depthwise_reflect(self, x, k2d: np.ndarray):
        self._f32(x)
        n, c, h, w = x.shape
        k = np.ascontiguousarray(k2d, dtype=np.float32)
        out = torch.empty_like(x)
        _check(lib().ss4k_op_depthwise_reflect(self._h, x.data_ptr(), out.data_ptr(), n * c, h, w,
                                               k.ctypes.data_as(C.c_void_p), k.shape[0], _stream()))
        torch.cuda.current_stream().synchronize()  # k lives on the host until the copy is done
        return out

    def plane_stats(self, x):
        self._f32(x)
        n, c, h, w = x.shape
        out = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
        _check(lib().ss4k_op_plane_stats(self._h, x.data_ptr(), out.data_ptr(), n * c, h * w, _stream()))
        return out

    def f32nchw_to_u8nhwc(self, x):
        self._f32(x)
        n, c, h, w = x.shape
        out = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
        _check(lib().ss4k_op_f32nchw_to_u8nhwc(self._h, x.data_ptr(), out.data_ptr(), n, c, h, w, _stream()))
        return out


def make_desc(kind: int, dtype: int = F32, scale: int = 2, num_feat: int = 64, num_block: int = 23,
              num_grow_ch: int = 32, bsvd_chns: Sequence[int] = (32, 64, 128), bsvd_mid_ch: int = 32,
              bsvd_interm_ch: int = 30, bsvd_stream: bool = False, flags: int = 0) -> ModelDesc:
    d = ModelDesc()
    d.kind, d.dtype, d.scale, d.num_feat, d.num_block, d.num_grow_ch = kind, dtype, scale, num_feat, num_block, num_grow_ch
    d.bsvd_chns = (C.c_int32 * 3)(*bsvd_chns)
    d.bsvd_mid_ch, d.bsvd_interm_ch = bsvd_mid_ch, bsvd_interm_ch
    d.bsvd_stream = 1 if bsvd_stream else 0
    d.flags = int(flags)
    return d


def param_count(desc: ModelDesc) -> int:
    return int(lib().ss4k_model_param_count(C.byref(desc)))


class Model:
    """A network resident on one GPU; callable like the reference's ``self.model`` (NCHW float in/out)."""

    def __init__(self, ctx: Context, desc: ModelDesc, flat_weights: np.ndarray):
        self.ctx, self.desc = ctx, desc
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        h = C.c_void_p()
        with torch.cuda.device(ctx.device):
            _check(lib().ss4k_model_create(ctx._h, C.byref(desc), w.ctypes.data_as(C.c_void_p), w.size, C.byref(h)))
        self._h = h
        self.in_channels = lib().ss4k_model_in_channels(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().ss4k_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, wait: bool = True) -> None:
        """Raise if an earlier forward of this model failed asynchronously (ss4k_model_check; only the chain kernel can)."""
        _check(lib().ss4k_model_check(self._h, 1 if wait else 0))

    def workspace_bytes(self, n, h, w) -> int:
        b = C.c_size_t()
        _check(lib().ss4k_model_workspace_bytes(self._h, n, h, w, C.byref(b)))
        return int(b.value)

    def out_shape(self, n, h, w):
        oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
        _check(lib().ss4k_model_out_shape(self._h, n, h, w, C.byref(oc), C.byref(oh), C.byref(ow)))
        return oc.value, oh.value, ow.value

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        squeeze_f = False
        seq_shape = None
        if x.ndim == 5:  # BSVD's (N, F, C, H, W)
            if self.desc.bsvd_stream:  # BSVD.forward flattens N*F into one stream (bsvd/model.py:520-522)
                seq_shape = x.shape[:2]
                x = x.reshape((-1,) + tuple(x.shape[2:]))
            else:
                assert x.shape[1] == 1, "per-frame BSVD takes F = 1 (fsrcnn_upscaler.py:277); build it with stream=True for F > 1"
                x = x[:, 0]
                squeeze_f = True
        assert x.ndim == 4 and x.shape[1] == self.in_channels, f"expected (N,{self.in_channels},H,W), got {tuple(x.shape)}"
        x = x.to(device=self.ctx.device, dtype=torch.float32).contiguous()
        n, _, h, w = x.shape
        oc, oh, ow = self.out_shape(n, h, w)
        out = torch.empty((n, oc, oh, ow), dtype=torch.float32, device=x.device)
        with torch.cuda.device(self.ctx.device):
            _check(lib().ss4k_model_forward(self._h, x.data_ptr(), out.data_ptr(), n, h, w, _stream()))
        if seq_shape is not None:
            return out.reshape(tuple(seq_shape) + tuple(out.shape[1:]))
        return out.unsqueeze(1) if squeeze_f else out

    forward = __call__

    def eval(self):
        return self


class Upscaler:
    """ss4k_upscaler: the frame-in/frame-out path (uint8 NHWC -> uint8 NHWC)."""

    def __init__(self, ctx: Context, sr: Model, lr_shape, output_shape=None, lr_hr_resize=True, single_mode=False,
                 denoise: Optional[Model] = None, denoise_rate: float = 1.0):
        self.ctx, self.sr, self.denoise = ctx, sr, denoise
        cfg = UpscaleCfg()
        cfg.lr_h, cfg.lr_w = int(lr_shape[0]), int(lr_shape[1])
        cfg.out_h, cfg.out_w = (0, 0) if output_shape is None else (int(output_shape[0]), int(output_shape[1]))
        cfg.lr_hr_resize, cfg.single_mode = int(bool(lr_hr_resize)), int(bool(single_mode))
        cfg.sr_is_realesrgan = int(sr.in_channels == 3)
        cfg.denoising, cfg.denoise_rate = int(denoise is not None), float(denoise_rate)
        h = C.c_void_p()
        _check(lib().ss4k_upscaler_create(ctx._h, C.byref(cfg), sr._h, denoise._h if denoise is not None else None,
                                          C.byref(h)))
        self._h, self.cfg = h, cfg

    def close(self):
        if getattr(self, "_h", None):
            lib().ss4k_upscaler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _check(lib().ss4k_upscaler_reset(self._h))

    def last_enqueue_ms(self):
        """(denoise_ms, model_ms): host time the last job spent enqueueing those stages."""
        d, m = C.c_double(), C.c_double()
        _check(lib().ss4k_upscaler_last_enqueue_ms(self._h, C.byref(d), C.byref(m)))
        return d.value, m.value

    def out_shape(self, n, h, w):
        oh, ow = C.c_int(), C.c_int()
        _check(lib().ss4k_upscaler_out_shape(self._h, n, h, w, C.byref(oh), C.byref(ow)))
        return oh.value, ow.value

    def enable_taps(self, on=True):
        _check(lib().ss4k_upscaler_enable_taps(self._h, int(on)))

    def read_tap(self, which: int) -> torch.Tensor:
        dims = (C.c_int * 4)()
        _check(lib().ss4k_upscaler_read_tap(self._h, which, None, 0, C.byref(dims), _stream()))
        out = torch.empty(tuple(dims), dtype=torch.float32, device=self.ctx.device)
        _check(lib().ss4k_upscaler_read_tap(self._h, which, out.data_ptr(), out.numel(), C.byref(dims), _stream()))
        return out

    def __call__(self, frames: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.ndim == 4 and frames.shape[-1] == 3
        frames = frames.contiguous()
        n, h, w, _ = frames.shape
        oh, ow = self.out_shape(n, h, w)
        if out is None:
            out = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=frames.device)
        with torch.cuda.device(self.ctx.device):
            _check(lib().ss4k_upscale_frames(self._h, frames.data_ptr(), n, h, w, out.data_ptr(), out.numel(), _stream()))
        return out
