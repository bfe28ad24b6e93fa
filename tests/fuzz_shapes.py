"""Test infrastructure (long-running, run by hand on a GPU box): random-shape parity fuzz of the conv networks (fp32 HIP vs the CPU oracle, rtol 1e-3 / atol 1e-4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
from oracle import nets as onets
from tests.helpers import assert_close, psnr

ctx = _capi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
bad = 0
rr_tab = {s: W.rrdbnet_table(11 + s, scale=s, num_feat=64, num_block=1, num_grow_ch=32) for s in (1, 2, 4)}
rr = {(s, d): factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=rr_tab[s], dtype=d, scale=s, num_block=1)
      for s in (1, 2, 4) for d in ("f32", "f16")}
sv_tab = W.srvgg_table(5, num_feat=32, num_conv=3, upscale=2)
sv = {d: _capi.Model(ctx, _capi.make_desc(_capi.SRVGG, _capi.F32 if d == "f32" else _capi.F16, scale=2, num_feat=32, num_block=3),
                     W.flatten(sv_tab, W.srvgg_keys(3))) for d in ("f32", "f16")}
bs_tab = W.bsvd_table(seed=21)
bs = {d: factory.build_denoise_model(ctx, weights=bs_tab, dtype=d) for d in ("f32", "f16")}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    kind = rng.choice(["rrdb", "srvgg", "bsvd"])
    n = int(rng.integers(1, 4))
    if kind == "rrdb":
        s = int(rng.choice([1, 2, 4])); r = {1: 4, 2: 2, 4: 1}[s]
        h, w = int(rng.integers(1, 30)) * r, int(rng.integers(1, 50)) * r
        x = torch.rand(n, 3, h, w)
        with torch.no_grad(): want = onets.rrdbnet(x, rr_tab[s], s, 1)
        got32, got16 = rr[(s, "f32")](x.cuda()), rr[(s, "f16")](x.cuda())
        name = f"rrdb x{s} {tuple(x.shape)}"
    elif kind == "srvgg":
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 120))
        x = torch.rand(n, 3, h, w)
        with torch.no_grad(): want = onets.srvgg(x, sv_tab, 3, 2)
        got32, got16 = sv["f32"](x.cuda()), sv["f16"](x.cuda())
        name = f"srvgg {tuple(x.shape)}"
    else:
        h, w = int(rng.integers(1, 16)) * 4, int(rng.integers(1, 24)) * 4
        x = torch.rand(n, 1, 4, h, w); x[:, :, 3] = 0.05
        with torch.no_grad(): want = onets.bsvd_f1(x, bs_tab)
        got32, got16 = bs["f32"](x.cuda()), bs["f16"](x.cuda())
        name = f"bsvd {tuple(x.shape)}"
    try:
        assert_close(got32, want, what=name)
        p16 = psnr(got16, want)
        assert p16 > 40, f"{name}: fp16 psnr {p16:.1f}"
        print("ok ", name, f"fp16 psnr {p16:.1f}", flush=True)
    except AssertionError as e:
        bad += 1
        print("BAD", name, str(e)[:200], flush=True)
print("failures:", bad)
