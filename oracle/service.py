"""Oracle for the service glue around the SR networks.  TEST INFRASTRUCTURE ONLY.

Restates ``FsrcnnUpscalerService.upscale / upscale_multi / upscale_single``
(reference ``src/upscale/fsrcnn_upscaler.py:144-326``) in PyTorch-CPU fp32, including the
quirks listed in SURVEY.md §8(a) (always-bicubic output resize, uint8 truncation, Bessel std
with eps added to sigma, first-frame noise map 0.05).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Tuple

import torch
import torch.nn.functional as F

LR_SHAPES = [(360, 640), (540, 960), (630, 1120), (720, 1280), (900, 1600), (1080, 1920)]


def gaussian_kernel2d(kernel_size: int, sigma: float) -> torch.Tensor:
    """Normalised 2-D gaussian, reference ``blur_ker`` (fsrcnn_upscaler.py:20-52)."""
    ax = torch.arange(kernel_size, dtype=torch.float32)
    xx = ax.repeat(kernel_size).view(kernel_size, kernel_size)
    yy = xx.t()
    mean = (kernel_size - 1) / 2.0
    var = sigma ** 2.0
    k = (1.0 / (2.0 * math.pi * var)) * torch.exp(-((xx - mean) ** 2.0 + (yy - mean) ** 2.0) / (2 * var))
    return k / torch.sum(k)


def sharpen_kernel2d(strength: float) -> torch.Tensor:
    """3x3 sharpen, reference ``sharpen_ker`` (fsrcnn_upscaler.py:54-84): centre 1+8s, rest -s."""
    sharp = torch.tensor([[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]])
    ident = torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0]])
    k = sharp * strength + (1 - strength) * ident
    return (k / torch.sum(k)).to(torch.float32)


def depthwise_reflect(x: torch.Tensor, k2d: torch.Tensor) -> torch.Tensor:
    """Single-channel conv with ``padding_mode='reflect'`` applied to every (n,c) plane."""
    n, c, h, w = x.shape
    p = k2d.shape[-1] // 2
    xp = F.pad(x.reshape(n * c, 1, h, w), (p, p, p, p), mode="reflect")
    return F.conv2d(xp, k2d.view(1, 1, *k2d.shape)).reshape(n, c, h, w)


def channel_match(hr: torch.Tensor, lr: torch.Tensor) -> torch.Tensor:
    """Per (n,c) mean / unbiased-std matching (fsrcnn_upscaler.py:188-199, :302-313)."""
    n, c, h, w = hr.shape
    hm = hr.reshape(n, c, -1).mean(-1).view(n, c, 1, 1)
    hs = hr.reshape(n, c, -1).std(-1).view(n, c, 1, 1)
    lm = lr.reshape(n, c, -1).mean(-1).view(n, c, 1, 1)
    ls = lr.reshape(n, c, -1).std(-1).view(n, c, 1, 1)
    return (hr - hm) / (hs + 1e-8) * ls + lm


def local_color_match(hr: torch.Tensor, lr: torch.Tensor, match_k: torch.Tensor) -> torch.Tensor:
    """Low-frequency colour match (fsrcnn_upscaler.py:201-218)."""
    n, c, h, w = hr.shape
    mf = 8
    if (h // mf) > (match_k.shape[-1] // 2) and h > 64 and w > 64:
        lb = F.interpolate(lr, size=(h // mf, w // mf), mode="area")
        hb = F.interpolate(hr, size=(h // mf, w // mf), mode="area")
        diff = depthwise_reflect(hb, match_k) - depthwise_reflect(lb, match_k)
        hr = hr - F.interpolate(diff, size=(h, w), mode="bilinear")
    return hr


class OracleUpscaler:
    """Same constructor arguments and ``upscale`` semantics as the reference service
    (``fsrcnn_upscaler.py:89-116``) with the model callables injected."""

    def __init__(self, sr_model: Callable, lr_level: int = 3, denoising: bool = False,
                 denoise_rate: float = 1.0, upscaler_model: str = "realesrgan",
                 lr_hr_resize: bool = True, denoise_model: Optional[Callable] = None,
                 output_shape: Optional[Tuple[int, int]] = None, single_mode: Optional[bool] = None,
                 lr_shape: Optional[Tuple[int, int]] = None):
        self.lr_shape = tuple(lr_shape) if lr_shape is not None else LR_SHAPES[lr_level]
        self.denoise_rate = denoise_rate
        self.output_shape = output_shape
        self.upscaler_model = upscaler_model
        self.single_mode = (upscaler_model != "realesrgan") if single_mode is None else single_mode
        self.denoising = denoising
        self.lr_hr_resize = lr_hr_resize
        self.model = sr_model
        self.denoise_model = denoise_model
        self.sharpen = sharpen_kernel2d(0.00002)
        self.sharpen_hr = sharpen_kernel2d(0.00007)
        self.match_k = gaussian_kernel2d(17, 8.0)
        self.first_frame = True

    # fsrcnn_upscaler.py:144-166
    def upscale(self, frames: torch.Tensor) -> torch.Tensor:
        assert frames.ndim == 4 and frames.shape[-1] == 3
        if self.single_mode:
            return torch.stack([self.upscale_single(frames[i]) for i in range(frames.shape[0])], 0)
        return self.upscale_multi(frames)

    def _resize_out(self, x: torch.Tensor) -> torch.Tensor:
        # quirk: the area/bicubic switch compares against N (or C) so bicubic always wins
        return F.interpolate(x, size=tuple(self.output_shape), mode="bicubic")

    # fsrcnn_upscaler.py:168-233
    def upscale_multi(self, frames: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        with torch.no_grad():
            img = frames.permute(0, 3, 1, 2) / 255.0
            lr = img
            if (img.shape[-1] > self.lr_shape[-1] or img.shape[-2] > self.lr_shape[-2]) and self.lr_hr_resize:
                lr = F.interpolate(img, size=self.lr_shape, mode="area")
            hr = self.model(lr)
            if taps is not None:
                taps["lr"] = lr.clone(); taps["model"] = hr.clone()
            hr = channel_match(hr, lr)
            if taps is not None:
                taps["stats"] = hr.clone()
            hr = local_color_match(hr, lr, self.match_k)
            if taps is not None:
                taps["color"] = hr.clone()
            hr = torch.clamp(hr, 0, 1)
            if self.output_shape is not None and self.lr_hr_resize:
                hr = self._resize_out(hr)
            hr = torch.clamp(hr, 0, 1)
            if taps is not None:
                taps["final"] = hr.clone()
            return (hr * 255).permute(0, 2, 3, 1).to(torch.uint8)

    # fsrcnn_upscaler.py:235-326
    def upscale_single(self, img: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        with torch.no_grad():
            x = img.permute(2, 0, 1).unsqueeze(0) / 255.0
            lr_before = lr = F.interpolate(x, size=self.lr_shape, mode="area").squeeze(0)  # (3,H,W)
            if self.denoising:
                c, h, w = lr.shape
                noise = 0.05 if self.first_frame else 0.1 * self.denoise_rate
                self.first_frame = False
                inp = torch.empty((1, 1, 4, h, w), dtype=torch.float32)
                inp[0, 0, :3] = lr
                inp[0, 0, 3] = noise
                den = self.denoise_model(inp)[:, -1].squeeze(0)  # (3,H,W)
                if taps is not None:
                    taps["denoise"] = den.clone()
                den = torch.clamp(depthwise_reflect(den.view(c, 1, h, w), self.sharpen).view(c, h, w), 0, 1)
                lr = den * 0.8 + (1 - 0.8) * lr
            if taps is not None:
                taps["lr"] = lr.clone()
            lr4 = lr.unsqueeze(1)  # (3,1,H,W)
            if self.upscaler_model == "realesrgan":
                hr = self.model(lr4.permute(1, 0, 2, 3).float()).permute(1, 0, 2, 3)
            else:
                hr = self.model(lr4.float())
            if self.denoising:
                hr = torch.clamp(depthwise_reflect(hr, self.sharpen_hr), 0, 1)
            if taps is not None:
                taps["model"] = hr.clone()
            c, f, h, w = hr.shape
            hr = channel_match(hr.view(1, c, h, w), lr_before.unsqueeze(0)).view(c, f, h, w)
            hr = torch.clamp(hr, 0, 1)
            if self.output_shape is not None:
                hr = self._resize_out(hr)
            hr = torch.clamp(hr, 0, 1)
            if taps is not None:
                taps["final"] = hr.clone()
            return (hr * 255)[:, 0].permute(1, 2, 0).to(torch.uint8)
