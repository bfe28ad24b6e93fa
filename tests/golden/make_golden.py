#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Container-only: needs /root/reference (absent on the GPU box).  It imports the reference's
own Python modules (nothing is copied), feeds them deterministic weights from
``sharkshark-4k_amd/weights.py`` plus seeded inputs, and stores inputs + outputs as small
``.npz`` fixtures.  It also checks the oracle (``oracle/``) against every vector and records the
max|delta| in MANIFEST.json, which is what "pinned" means in oracle/__init__.py.

Shims (all confined to this process):
  * stub modules for imports the image lacks: cv2, basicsr.archs.rrdbnet_arch,
    basicsr.utils.download_util, realesrgan (module-level imports at
    realesrgan/factory.py:6-9, fsrcnn_upscaler.py:1);
  * BSVD builds ``torch.zeros(..., device='cuda')`` and calls ``.cuda()``
    (bsvd/model.py:87,108,123,545): on this CPU-only box those are redirected to CPU.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import sharkshark4k_amd  # noqa: E402  (alias loader for the hyphenated package dir)
from sharkshark4k_amd import weights as W  # noqa: E402
from oracle import nets as onets  # noqa: E402
from oracle import service as osvc  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


# ------------------------------------------------------------------ shims
def _install_shims():
    for name in ("cv2", "basicsr", "basicsr.archs", "basicsr.archs.rrdbnet_arch", "basicsr.utils",
                 "basicsr.utils.download_util", "realesrgan"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["basicsr.archs.rrdbnet_arch"].RRDBNet = object
    sys.modules["basicsr.utils.download_util"].load_file_from_url = lambda *a, **k: None
    sys.modules["realesrgan"].RealESRGANer = object
    _zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return _zeros(*a, **k)

    torch.zeros = zeros_cpu
    torch.Tensor.cuda = lambda self, *a, **k: self


_install_shims()
import src.upscale.model.fsrcnn.model as ref_fsrcnn  # noqa: E402
import src.upscale.model.realesrgan.factory as ref_esr  # noqa: E402
import src.upscale.model.bsvd.model as ref_bsvd  # noqa: E402
import src.upscale.fsrcnn_upscaler as ref_svc  # noqa: E402
from src.util.profiler import Profiler as RefProfiler  # noqa: E402
from src.upscale.upscaler_base import UpscalerQueueEntry as RefEntry  # noqa: E402


def tt(table):
    return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in table.items())


def rng_u8(seed, shape):
    return np.random.default_rng(seed).integers(0, 256, size=shape, dtype=np.uint8)


def smooth_u8(seed, shape):
    """Box-blurred noise: natural-image-like low frequencies so PSNR/colour match are meaningful."""
    n, h, w, c = shape
    g = np.random.default_rng(seed).random((n, h + 8, w + 8, c)).astype(np.float32)
    k = 9
    cs = np.cumsum(np.cumsum(np.pad(g, ((0, 0), (1, 0), (1, 0), (0, 0))), 1), 2)
    box = (cs[:, k:, k:] - cs[:, :-k, k:] - cs[:, k:, :-k] + cs[:, :-k, :-k]) / (k * k)
    box = (box - box.min()) / (box.max() - box.min())
    return (box[:, :h, :w] * 255).astype(np.uint8)


manifest = {}


def save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    meta = dict(meta)
    meta["bytes"] = os.path.getsize(path)
    manifest[name] = meta
    print(f"{name}: {meta}")


def maxdiff(a, b):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else b
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


# ------------------------------------------------------------------ FSRCNN
def build_ref_fsrcnn(table, factor):
    m = ref_fsrcnn.FSRCNN(factor).eval()
    m.load_state_dict(tt(table))
    return m


def fsrcnn_cases():
    for factor in (2, 4):
        ck = torch.load(f"{REF}/src/upscale/model/fsrcnn/fsrcnn_x{factor}-T91.pth", map_location="cpu",
                        weights_only=False)["state_dict"]
        t91 = OrderedDict((k, ck[k].numpy().astype(np.float32)) for k in W.fsrcnn_keys())
        assert list(ck.keys()) == W.fsrcnn_keys(), "state_dict key order differs from weights.fsrcnn_keys()"
        np.save(os.path.join(HERE, f"fsrcnn_x{factor}_T91_flat.npy"), W.flatten(t91, W.fsrcnn_keys()))
        for tag, table in (("t91", t91), ("gen", W.fsrcnn_table(seed=factor))):
            x = np.random.default_rng(10 + factor).random((3, 1, 40, 56), dtype=np.float32)
            with torch.no_grad():
                y = build_ref_fsrcnn(table, factor)(torch.from_numpy(x))
                yo = onets.fsrcnn(torch.from_numpy(x), table, factor)
            save(f"fsrcnn_x{factor}_{tag}", {"oracle_maxdiff": maxdiff(y, yo), "factor": factor,
                                             "weights": "T91 checkpoint" if tag == "t91" else f"fsrcnn_table(seed={factor})"},
                 x=x, y=y.numpy())


# ------------------------------------------------------------------ SRVGG
SRVGG_CASES = [("srvgg_f64_c4_x4", 64, 4, 4, 11), ("srvgg_f32_c2_x2", 32, 2, 2, 12), ("srvgg_f16_c2_x4", 16, 2, 4, 13)]
# the reference's SHIPPED DEFAULT at its full depth (realesr-general-x4v3: num_feat 64, num_conv 32, x4; realesrgan/factory.py:88,132-138):
# plain generated weights, and a DNI blend (factory.py:152-157) whose PReLU slopes leave [0, 1] (both forms of the HIP epilogue)
SRVGG_FULL_CASES = [("srvgg_f64_c32_x4", 14, None), ("srvgg_f64_c32_x4_dni_wild", 15, 16)]


def build_ref_srvgg(table, nf, nc, up):
    m = ref_esr.SRVGGNetCompact(num_in_ch=3, num_out_ch=3, num_feat=nf, num_conv=nc, upscale=up, act_type="prelu").eval()
    m.load_state_dict(tt(table))
    return m


def srvgg_cases():
    for name, nf, nc, up, seed in SRVGG_CASES:
        table = W.srvgg_table(seed=seed, num_feat=nf, num_conv=nc, upscale=up)
        x = np.random.default_rng(seed).random((2, 3, 24, 40), dtype=np.float32)
        with torch.no_grad():
            y = build_ref_srvgg(table, nf, nc, up)(torch.from_numpy(x))
            yo = onets.srvgg(torch.from_numpy(x), table, nc, up)
        save(name, {"oracle_maxdiff": maxdiff(y, yo), "num_feat": nf, "num_conv": nc, "upscale": up,
                    "weights": f"srvgg_table(seed={seed})"}, x=x, y=y.numpy())


def srvgg_full_depth_cases():
    from tests.helpers import srvgg_full_table
    for name, seed, seed_b in SRVGG_FULL_CASES:
        table = srvgg_full_table(seed, seed_b)
        x = np.random.default_rng(seed).random((1, 3, 22, 38), dtype=np.float32)
        with torch.no_grad():
            y = build_ref_srvgg(table, 64, 32, 4)(torch.from_numpy(x))
            yo = onets.srvgg(torch.from_numpy(x), table, 32, 4)
        save(name, {"oracle_maxdiff": maxdiff(y, yo), "num_feat": 64, "num_conv": 32, "upscale": 4,
                    "weights": f"tests.helpers.srvgg_full_table({seed}, {seed_b})"}, x=x, y=y.numpy())
    # ... and through the reference SERVICE on the default configuration's path: batched mode, area pre-resize to lr_shape,
    # x4 network, statistics + local colour match, the always-bicubic resize to output_shape (fsrcnn_upscaler.py:168-233)
    import warnings
    warnings.filterwarnings("ignore")
    table = srvgg_full_table(15, 16)
    frames = smooth_u8(49, (1, 90, 140, 3))
    lr_shape, out_shape = (72, 104), (144, 208)
    svc = make_ref_service(build_ref_srvgg(table, 64, 32, 4), "realesrgan", False, None, lr_shape, out_shape, True, 1.0, None)
    osv = osvc.OracleUpscaler(lambda x: onets.srvgg(x, table, 32, 4), upscaler_model="realesrgan", output_shape=out_shape, lr_shape=lr_shape)
    ft = torch.from_numpy(frames)
    out1, out2 = svc.upscale(ft).numpy(), svc.upscale(ft).numpy()
    d = max(maxdiff(out1, osv.upscale(ft).numpy()), maxdiff(out2, osv.upscale(ft).numpy()))
    save("svc_multi_srvgg64x32_x4_area_bicubic", {"oracle_maxdiff_u8": d, "mode": "realesrgan", "sr": "srvgg",
                                                 "num_feat": 64, "num_conv": 32, "upscale": 4, "weights": "tests.helpers.srvgg_full_table(15, 16)",
                                                 "lr_shape": list(lr_shape), "output_shape": list(out_shape), "lr_hr_resize": True,
                                                 "denoising": False, "denoise_rate": 1.0, "single_mode": None, "bsvd_seed": 21},
         frames=frames, out1=out1, out2=out2)


# ------------------------------------------------------------------ BSVD
def build_ref_bsvd(table):
    m = ref_bsvd.BSVD(chns=[32, 64, 128], mid_ch=32, shift_input=False, norm="none", interm_ch=30,
                      act="relu6", pretrain_ckpt=None).eval()
    missing = m.load_state_dict(tt(table), strict=True)
    return m


def bsvd_cases():
    table = W.bsvd_table(seed=21)
    ref = build_ref_bsvd(table)
    for name, h, w, seed in (("bsvd32_f1_32x48", 32, 48, 22), ("bsvd32_f1_24x40", 24, 40, 23)):
        x = np.random.default_rng(seed).random((1, 1, 4, h, w), dtype=np.float32)
        x[:, :, 3] = 0.05
        with torch.no_grad():
            y = ref(torch.from_numpy(x.copy()))
            y2 = ref(torch.from_numpy(x.copy()))  # streaming state must reset (bsvd/model.py:579)
            yo = onets.bsvd_f1(torch.from_numpy(x.copy()), table)
        assert maxdiff(y, y2) == 0.0
        save(name, {"oracle_maxdiff": maxdiff(y, yo), "weights": "bsvd_table(seed=21)"}, x=x, y=y.numpy())
    # multi-frame streams through the bidirectional buffers (bsvd/model.py:515-580)
    for name, n, f, h, w, seed in (("bsvd32_seq5_24x40", 1, 5, 24, 40, 24), ("bsvd32_seq3_16x24", 1, 3, 16, 24, 25),
                                   ("bsvd32_seq2x2_16x24", 2, 2, 16, 24, 26)):
        x = np.random.default_rng(seed).random((n, f, 4, h, w), dtype=np.float32)
        x[:, :, 3] = 0.05
        with torch.no_grad():
            y = ref(torch.from_numpy(x.copy()))
            yo = onets.bsvd_seq(torch.from_numpy(x.copy()), table)
        save(name, {"oracle_maxdiff": maxdiff(y, yo), "weights": "bsvd_table(seed=21)", "stream": True}, x=x, y=y.numpy())


# ------------------------------------------------------------------ service glue
def make_ref_service(model, upscaler_model, denoising, denoise_model, lr_shape, output_shape, lr_hr_resize,
                     denoise_rate, single_mode=None):
    svc = ref_svc.FsrcnnUpscalerService(lr_level=3, device="cpu", denoising=denoising, denoise_rate=denoise_rate,
                                        upscaler_model=upscaler_model, batch_size=1, jit_mode=False,
                                        lr_hr_resize=lr_hr_resize)
    svc.lr_shape = lr_shape
    svc.output_shape = output_shape
    if single_mode is not None:
        svc.single_mode = single_mode
    # what proc_init() would have built (fsrcnn_upscaler.py:118-139), in fp32 on CPU
    svc.lr_prev = None
    svc.model = model
    if denoising:
        svc.denoise_model = denoise_model
        svc.denoise_blur = ref_svc.blur_ker()
        svc.denoise_sharpen = ref_svc.sharpen_ker(strength=0.00002)
        svc.denoise_sharpen_hr = ref_svc.sharpen_ker(strength=0.00007)
    svc.match_blur = ref_svc.blur_ker(kernel_size=8 * 2 + 1, sigma=8.0)
    svc.profiler = RefProfiler()
    return svc


def service_cases():
    import warnings
    warnings.filterwarnings("ignore")
    srv_t = W.srvgg_table(seed=31, num_feat=32, num_conv=2, upscale=4)
    srv2_t = W.srvgg_table(seed=32, num_feat=32, num_conv=2, upscale=2)
    fs2_t = W.fsrcnn_table(seed=2)
    fs4_t = W.fsrcnn_table(seed=4)
    bs_t = W.bsvd_table(seed=21)
    ref_srv = build_ref_srvgg(srv_t, 32, 2, 4)
    ref_srv2 = build_ref_srvgg(srv2_t, 32, 2, 2)
    ref_bs = build_ref_bsvd(bs_t)

    cases = [
        # name, mode, sr, frames, lr_shape, output_shape, lr_hr_resize, denoising, denoise_rate, single_mode
        ("svc_multi_srvgg_x4_color", "realesrgan", ("srvgg", 31, 32, 2, 4), smooth_u8(41, (2, 24, 32, 3)), (24, 32), None, True, False, 1.0, None),
        ("svc_multi_srvgg_x4_area_bicubic", "realesrgan", ("srvgg", 31, 32, 2, 4), smooth_u8(42, (2, 70, 98, 3)), (24, 32), (60, 80), True, False, 1.0, None),
        ("svc_multi_srvgg_x2_nocolor", "realesrgan", ("srvgg", 32, 32, 2, 2), rng_u8(43, (1, 24, 40, 3)), (24, 40), None, True, False, 1.0, None),
        ("svc_multi_srvgg_x4_noresize_flag", "realesrgan", ("srvgg", 31, 32, 2, 4), smooth_u8(44, (1, 40, 56, 3)), (24, 32), (60, 80), False, False, 1.0, None),
        ("svc_single_fsrcnn_x2", "fsrcnn", ("fsrcnn", 2, 2), smooth_u8(45, (2, 36, 52, 3)), (36, 52), None, True, False, 1.0, None),
        ("svc_single_fsrcnn_x4_bicubic", "fsrcnn", ("fsrcnn", 4, 4), rng_u8(46, (2, 50, 70, 3)), (24, 36), (60, 100), True, False, 1.0, None),
        ("svc_single_fsrcnn_x2_denoise", "fsrcnn", ("fsrcnn", 2, 2), smooth_u8(47, (3, 32, 48, 3)), (32, 48), None, True, True, 0.7, None),
        ("svc_single_srvgg_x2_denoise", "realesrgan", ("srvgg", 32, 32, 2, 2), smooth_u8(48, (2, 32, 48, 3)), (32, 48), (80, 120), True, True, 1.0, True),
    ]
    for (name, mode, sr, frames, lr_shape, out_shape, lrhr, den, drate, single) in cases:
        if sr[0] == "srvgg":
            _, seed, nf, nc, up = sr
            table = srv_t if seed == 31 else srv2_t
            ref_model = ref_srv if seed == 31 else ref_srv2
            o_model = (lambda t, nc, up: (lambda x: onets.srvgg(x, t, nc, up)))(table, nc, up)
            sr_meta = {"sr": "srvgg", "seed": seed, "num_feat": nf, "num_conv": nc, "upscale": up}
        else:
            _, seed, factor = sr
            table = fs2_t if factor == 2 else fs4_t
            ref_model = build_ref_fsrcnn(table, factor)
            o_model = (lambda t, f: (lambda x: onets.fsrcnn(x, t, f)))(table, factor)
            sr_meta = {"sr": "fsrcnn", "seed": seed, "factor": factor}
        svc = make_ref_service(ref_model, mode, den, ref_bs, lr_shape, out_shape, lrhr, drate, single)
        osv = osvc.OracleUpscaler(o_model, denoising=den, denoise_rate=drate, upscaler_model=mode,
                                  lr_hr_resize=lrhr, denoise_model=lambda x: onets.bsvd_f1(x, bs_t),
                                  output_shape=out_shape, single_mode=single, lr_shape=lr_shape)
        # two consecutive jobs so the first-frame / later-frame noise-map branch is covered
        ft = torch.from_numpy(frames)
        entry = RefEntry(frames=ft, step=7, profiler=RefProfiler())
        res = svc.proc_job_recieved(entry)
        out1 = res.frames.numpy()
        out2 = svc.upscale(ft).numpy()
        o1 = osv.upscale(ft).numpy()
        o2 = osv.upscale(ft).numpy()
        d = max(maxdiff(out1, o1), maxdiff(out2, o2))
        assert res.step == 7 and "upscaler.upscale" in res.profiler.data
        save(name, {"oracle_maxdiff_u8": d, "mode": mode, **sr_meta, "lr_shape": list(lr_shape),
                    "output_shape": None if out_shape is None else list(out_shape), "lr_hr_resize": lrhr,
                    "denoising": den, "denoise_rate": drate, "single_mode": single,
                    "bsvd_seed": 21, "profiler_keys": sorted(res.profiler.data.keys())},
             frames=frames, out1=out1, out2=out2)


# ------------------------------------------------------------------ resampling known-answer vectors
def kat_cases():
    g = np.random.default_rng(51)
    x = g.random((2, 3, 23, 37), dtype=np.float32)
    xt = torch.from_numpy(x)
    import torch.nn.functional as F
    arrays = {"x": x}
    arrays["area_9x14"] = F.interpolate(xt, size=(9, 14), mode="area").numpy()
    arrays["area_23x37"] = F.interpolate(xt, size=(23, 37), mode="area").numpy()
    arrays["area_30x50"] = F.interpolate(xt, size=(30, 50), mode="area").numpy()
    arrays["bicubic_31x50"] = F.interpolate(xt, size=(31, 50), mode="bicubic").numpy()
    arrays["bicubic_11x19"] = F.interpolate(xt, size=(11, 19), mode="bicubic").numpy()
    arrays["bilinear_46x80"] = F.interpolate(xt, size=(46, 80), mode="bilinear").numpy()
    arrays["blur17"] = ref_svc.blur_ker(kernel_size=17, sigma=8.0)(xt.reshape(6, 1, 23, 37)).detach().numpy().reshape(2, 3, 23, 37)
    arrays["sharpen_hr"] = ref_svc.sharpen_ker(strength=0.00007)(xt.reshape(6, 1, 23, 37)).detach().numpy().reshape(2, 3, 23, 37)
    arrays["blur17_weight"] = ref_svc.blur_ker(kernel_size=17, sigma=8.0).weight.detach().numpy().reshape(17, 17)
    arrays["sharpen_weight"] = ref_svc.sharpen_ker(strength=0.00002).weight.detach().numpy().reshape(3, 3)
    d1 = maxdiff(arrays["blur17"], osvc.depthwise_reflect(xt, osvc.gaussian_kernel2d(17, 8.0)))
    d2 = maxdiff(arrays["sharpen_hr"], osvc.depthwise_reflect(xt, osvc.sharpen_kernel2d(0.00007)))
    d3 = maxdiff(arrays["blur17_weight"], osvc.gaussian_kernel2d(17, 8.0))
    d4 = maxdiff(arrays["sharpen_weight"], osvc.sharpen_kernel2d(0.00002))
    save("kat_resample", {"oracle_maxdiff": max(d1, d2, d3, d4)}, **arrays)


if __name__ == "__main__":
    fsrcnn_cases()
    srvgg_cases()
    bsvd_cases()
    service_cases()
    srvgg_full_depth_cases()
    kat_cases()
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "torch": torch.__version__,
                   "reference": "gmlwns2000/sharkshark-4k @ /root/reference", "cases": manifest}, f, indent=1)
    bad = {k: v for k, v in manifest.items() if max(v.get("oracle_maxdiff", 0), v.get("oracle_maxdiff_u8", 0)) > 1e-6}
    print("oracle mismatches:", bad)
    sys.exit(1 if bad else 0)
