"""Dev tool: random-shape fuzz of the routes.  RRDBNet on the 32x32x16 route with the fused dense-block pairs (SS4K_MODEL_NO_W16 |
NO_UPS_PRESUM) against the same with four launches per dense block (| NO_DENSE: conv1..conv4 on conv_mfma.hip's 32-cout tile) must agree BIT FOR
BIT - conv5 runs on the wide kernel with its residual through the matrix core on both sides since round 5; the default route
(conv_w16.hip for the 64-cout layers and conv5: another summation order) must sit within 6e-3 of the output peak and
60 dB of it; all under one and two launch chains.  usage: python tools/fuzz_routes.py [cases=150] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = _capi.Context(0)
models = {}
def model(scale, flags):
    k = (scale, flags)
    if k not in models:
        t = W.rrdbnet_table(40 + scale, scale=scale, num_block=1)
        models[k] = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=scale, num_block=1, flags=flags), W.flatten(t, W.rrdbnet_keys(1)))
    return models[k]
bad = 0
for i in range(cases):
    scale = int(rng.choice([1, 2, 4]))
    r = {1: 4, 2: 2, 4: 1}[scale]
    n = int(rng.integers(1, 5))
    gh, gw = int(rng.integers(1, 70)), int(rng.integers(1, 140))   # interior grid (what the fused kernels see)
    if rng.random() < 0.3: gh = int(rng.choice([15, 16, 17, 31, 32, 33, 48]))
    if rng.random() < 0.3: gw = int(rng.choice([30, 31, 32, 33, 34, 63, 64, 65, 96]))
    lanes = _capi.MODEL_TWO_CHAINS if (n % 2 == 0 and rng.random() < 0.5) else _capi.MODEL_ONE_CHAIN
    x = torch.rand(n, 3, gh * r, gw * r, generator=torch.Generator().manual_seed(i)).cuda()
    pin = _capi.MODEL_NO_W16 | _capi.MODEL_NO_UPS_PRESUM   # (the pre-summed up-sampling convs are not bit-identical either)
    want = model(scale, lanes | pin | _capi.MODEL_NO_DENSE)(x).clone()
    got = model(scale, lanes | pin)(x).clone()
    dflt = model(scale, lanes)(x)
    peak = float(want.abs().max()) + 1e-20
    mse = float(((dflt - want) / peak).double().pow(2).mean())
    ok = bool(torch.isfinite(got).all()) and torch.equal(got, want) and bool(torch.isfinite(dflt).all()) and \
        float((dflt - want).abs().max()) <= 6e-3 * peak and (mse == 0.0 or -10.0 * np.log10(mse) > 60.0)
    if not ok:
        bad += 1
        print(f"case {i}: scale {scale} n {n} grid {gh}x{gw} lanes {lanes}: MISMATCH max |d| {float((got - want).abs().max()):.3g}", flush=True)
print(f"{cases} cases, {bad} mismatching:", "FUZZ FAILED" if bad else "FUZZ OK")
