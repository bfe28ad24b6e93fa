"""CPU: the oracle reproduces every golden vector captured from the reference itself."""
import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import weights as W
from oracle import nets as onets
from oracle import service as osvc
from tests.conftest import load_golden, manifest
from tests.helpers import srvgg_table_for

CASES = manifest()


def _t91(factor):
    import os
    from tests.conftest import GOLDEN
    flat = np.load(os.path.join(GOLDEN, f"fsrcnn_x{factor}_T91_flat.npy"))
    table, pos = {}, 0
    ref = W.fsrcnn_table(0)
    for k in W.fsrcnn_keys():
        n = ref[k].size
        table[k] = flat[pos:pos + n].reshape(ref[k].shape)
        pos += n
    assert pos == flat.size == 12809
    return table


@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("tag", ["t91", "gen"])
def test_fsrcnn_matches_reference(factor, tag):
    g = load_golden(f"fsrcnn_x{factor}_{tag}")
    table = _t91(factor) if tag == "t91" else W.fsrcnn_table(seed=factor)
    with torch.no_grad():
        y = onets.fsrcnn(torch.from_numpy(g["x"]), table, factor).numpy()
    assert np.array_equal(y, g["y"])


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("srvgg_")])
def test_srvgg_matches_reference(name):
    m = CASES[name]
    g = load_golden(name)
    table = srvgg_table_for(m)
    with torch.no_grad():
        y = onets.srvgg(torch.from_numpy(g["x"]), table, m["num_conv"], m["upscale"]).numpy()
    assert np.array_equal(y, g["y"])


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("bsvd32_")])
def test_bsvd_f1_matches_reference(name):
    g = load_golden(name)
    with torch.no_grad():
        fn = onets.bsvd_seq if "_seq" in name else onets.bsvd_f1
        y = fn(torch.from_numpy(g["x"]), W.bsvd_table(seed=21)).numpy()
    assert np.array_equal(y, g["y"])


def oracle_service_from_manifest(m):
    if m["sr"] == "srvgg":
        t = srvgg_table_for(m)
        model = lambda x: onets.srvgg(x, t, m["num_conv"], m["upscale"])
    else:
        t = W.fsrcnn_table(seed=m["seed"])
        model = lambda x: onets.fsrcnn(x, t, m["factor"])
    bs = W.bsvd_table(seed=m["bsvd_seed"])
    return osvc.OracleUpscaler(model, denoising=m["denoising"], denoise_rate=m["denoise_rate"], upscaler_model=m["mode"],
                               lr_hr_resize=m["lr_hr_resize"], denoise_model=lambda x: onets.bsvd_f1(x, bs),
                               output_shape=None if m["output_shape"] is None else tuple(m["output_shape"]),
                               single_mode=m["single_mode"], lr_shape=tuple(m["lr_shape"]))


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("svc_")])
def test_service_glue_matches_reference(name):
    g = load_golden(name)
    svc = oracle_service_from_manifest(CASES[name])
    frames = torch.from_numpy(g["frames"])
    assert np.array_equal(svc.upscale(frames).numpy(), g["out1"])
    assert np.array_equal(svc.upscale(frames).numpy(), g["out2"])  # second job: later-frame noise map


def test_resample_known_answers():
    g = load_golden("kat_resample")
    x = torch.from_numpy(g["x"])
    assert np.array_equal(osvc.depthwise_reflect(x, osvc.gaussian_kernel2d(17, 8.0)).numpy(), g["blur17"])
    assert np.array_equal(osvc.depthwise_reflect(x, osvc.sharpen_kernel2d(0.00007)).numpy(), g["sharpen_hr"])
    assert np.array_equal(osvc.gaussian_kernel2d(17, 8.0).numpy(), g["blur17_weight"])
    assert np.array_equal(osvc.sharpen_kernel2d(0.00002).numpy(), g["sharpen_weight"])


def test_rrdbnet_self_checks():
    """RRDBNet is unpinned by the reference (basicsr absent): check structure only."""
    assert W.num_params(W.rrdbnet_table(0, scale=2)) == 16_703_171
    assert W.num_params(W.rrdbnet_table(0, scale=4)) == 16_697_987
    t = W.rrdbnet_table(3, scale=2, num_block=1)
    x = torch.rand(1, 3, 16, 24)
    with torch.no_grad():
        y = onets.rrdbnet(x, t, 2, 1)
    assert y.shape == (1, 3, 32, 48) and torch.isfinite(y).all()
    t4 = W.rrdbnet_table(3, scale=4, num_block=1)
    with torch.no_grad():
        assert onets.rrdbnet(x, t4, 4, 1).shape == (1, 3, 64, 96)


def test_param_counts():
    assert W.num_params(W.fsrcnn_table(0)) == 12_809
    assert W.num_params(W.srvgg_table(0)) == 1_213_296
    assert W.num_params(W.bsvd_table(0)) == 2_454_583


class _ModuleRRDBNet(torch.nn.Module):
    """A second, independent coding of the published BasicSR RRDBNet as an ``nn.Module`` whose
    ``state_dict`` uses the upstream key names (``body.{i}.rdb{j}.conv{k}``, ``conv_first`` ...), i.e.
    what a real RealESRGAN checkpoint loads into.  Cannot pin the oracle to the reference (BasicSR is
    not in the image) but catches a slip in either coding and in the key map."""

    class RDB(torch.nn.Module):
        def __init__(self, nf, gc):
            super().__init__()
            for k in range(1, 6):
                setattr(self, f"conv{k}", torch.nn.Conv2d(nf + (k - 1) * gc, gc if k < 5 else nf, 3, 1, 1))

        def forward(self, x):
            feats = [x]
            for k in range(1, 5):
                feats.append(torch.nn.functional.leaky_relu(getattr(self, f"conv{k}")(torch.cat(feats, 1)), 0.2))
            return self.conv5(torch.cat(feats, 1)) * 0.2 + x

    class RRDB(torch.nn.Module):
        def __init__(self, nf, gc):
            super().__init__()
            self.rdb1, self.rdb2, self.rdb3 = (_ModuleRRDBNet.RDB(nf, gc) for _ in range(3))

        def forward(self, x):
            return self.rdb3(self.rdb2(self.rdb1(x))) * 0.2 + x

    def __init__(self, scale, num_block, nf=64, gc=32):
        super().__init__()
        self.scale = scale
        self.conv_first = torch.nn.Conv2d(3 * {1: 16, 2: 4, 4: 1}[scale], nf, 3, 1, 1)
        self.body = torch.nn.Sequential(*[self.RRDB(nf, gc) for _ in range(num_block)])
        for name in ("conv_body", "conv_up1", "conv_up2", "conv_hr"):
            setattr(self, name, torch.nn.Conv2d(nf, nf, 3, 1, 1))
        self.conv_last = torch.nn.Conv2d(nf, 3, 3, 1, 1)

    def forward(self, x):
        F_ = torch.nn.functional
        if self.scale != 4:
            x = F_.pixel_unshuffle(x, 2 if self.scale == 2 else 4)
        feat = self.conv_first(x)
        feat = feat + self.conv_body(self.body(feat))
        feat = F_.leaky_relu(self.conv_up1(F_.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        feat = F_.leaky_relu(self.conv_up2(F_.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        return self.conv_last(F_.leaky_relu(self.conv_hr(feat), 0.2))


@pytest.mark.parametrize("scale", [1, 2, 4])
def test_rrdbnet_functional_oracle_matches_module_form(scale):
    table = W.rrdbnet_table(31 + scale, scale=scale, num_block=2)
    net = _ModuleRRDBNet(scale, 2).eval()
    missing = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in table.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(net.state_dict().keys()) == W.rrdbnet_keys(2)  # upstream key order = flat blob order
    x = torch.rand(1, 3, 16, 24)
    with torch.no_grad():
        a, b = net(x), onets.rrdbnet(x, table, scale, 2)
    assert a.shape == b.shape == (1, 3, 16 * scale, 24 * scale)
    assert torch.allclose(a, b, rtol=0, atol=1e-6)



@pytest.mark.parametrize("factor", [2, 4])
def test_fsrcnn_t91_activation_range(factor):
    """The HIP path's fp16 hi/lo-split stages need every operand inside the fp16 range (csrc/models.cpp, FSRCNN build).
    On the real T91 checkpoints every tensor those stages split - the shrink output, the four mapping outputs and the
    expand output - stays below 10^3 for image-range inputs (measured: 116) (flat black / white, noise, a checkerboard at the pixel
    pitch), two orders of magnitude inside 65504; the weights are below 11."""
    import torch.nn.functional as F
    w = _t91(factor)
    assert max(float(np.abs(v).max()) for v in w.values()) < 16.0
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(64), torch.arange(64), indexing="ij")
    inputs = [torch.zeros(1, 1, 64, 64), torch.ones(1, 1, 64, 64), torch.rand(1, 1, 64, 64, generator=g),
              ((yy + xx) % 2).float()[None, None], (torch.rand(1, 1, 64, 64, generator=g) > 0.5).float()]
    t = lambda k: torch.from_numpy(np.asarray(w[k]))
    worst = 0.0
    with torch.no_grad():
        for x in inputs:
            y = F.prelu(F.conv2d(x, t("feature_extraction.0.weight"), t("feature_extraction.0.bias"), padding=2), t("feature_extraction.1.weight"))
            y = F.prelu(F.conv2d(y, t("shrink.0.weight"), t("shrink.0.bias")), t("shrink.1.weight"))
            worst = max(worst, float(y.abs().max()))
            for i in range(4):
                y = F.prelu(F.conv2d(y, t(f"map.{2*i}.weight"), t(f"map.{2*i}.bias"), padding=1), t(f"map.{2*i+1}.weight"))
                worst = max(worst, float(y.abs().max()))
            y = F.prelu(F.conv2d(y, t("expand.0.weight"), t("expand.0.bias")), t("expand.1.weight"))
            worst = max(worst, float(y.abs().max()))
    assert worst < 1000.0, worst
