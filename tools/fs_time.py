"""Dev tool: FSRCNN x2 / x4 on 12 planes of 720p: ms per call with the split-precision fp16 MFMA tail and with the
exact-fp32 kernels (SS4K_FS_EXACT=1 is read when a model is built: run the tool both ways), and the difference
between the two on the real T91 checkpoint values."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
for factor in (2, 4):
    m = factory.build_model_fsrcnn(ctx, factor=factor, weights=W.fsrcnn_table(seed=2))
    x = torch.rand(12, 1, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
    for _ in range(2): m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): m(x)
    e1.record(); torch.cuda.synchronize()
    print("fsrcnn x%d 12 planes 720p (%s tail): %.2f ms" % (factor, "exact fp32" if os.environ.get("SS4K_FS_EXACT") == "1" else "split fp16", e0.elapsed_time(e1) / 5), flush=True)
    if len(sys.argv) > 1:
        y = m(x[:3, :, :256, :384]).cpu().numpy()
        f = sys.argv[1] + f"_x{factor}.npy"
        if os.path.exists(f):
            r = np.load(f); d = np.abs(y - r)
            print(f"   vs {f}: max |diff| {d.max():.3e}, output peak {np.abs(r).max():.3f}, max rel-to-peak {d.max() / np.abs(r).max():.2e}")
        else:
            np.save(f, y)
