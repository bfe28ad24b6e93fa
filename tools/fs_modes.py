"""Dev tool: FSRCNN x2 on 12 planes of 720p in its three arithmetic modes - fp32-accurate split (default), exact fp32
(SS4K_MODEL_FS_EXACT), plain fp16 operands (dtype f16) - ms per call, ms per stage (ss4k_prof_read_kind) and the difference
of every mode's output from the exact kernels on a crop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
tab = W.fsrcnn_table(seed=2)
x = torch.rand(12, 1, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
ref = None
for name, kw in (("exact", dict(flags=_capi.MODEL_FS_EXACT)), ("split", {}), ("f16", dict(dtype="f16"))):
    m = factory.build_model_fsrcnn(ctx, factor=2, weights=tab, **kw)
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    ctx.prof_reset(); ctx.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m(x)
    e1.record(); torch.cuda.synchronize()
    st = {k: ctx.prof_read_kind(kid) for k, kid in (("head", 1), ("map", 2), ("tail", 3))}
    ctx.prof_enable(False)
    y = m(x[:3, :, :256, :384]).float().cpu()
    if ref is None: ref = y
    d = (y - ref).abs()
    mse = float(((y - ref) ** 2).mean())
    psnr = float("inf") if mse == 0 else 10 * torch.log10(torch.tensor(1.0 / mse)).item()
    print(f"{name:6s} {e0.elapsed_time(e1) / 10:.3f} ms/call  stages " + " ".join(f"{k} {v[1] / max(1, v[0]):.3f}" for k, v in st.items())
          + f"  | vs exact: max |d| {float(d.max()):.2e} (peak {float(ref.abs().max()):.2f}) PSNR {psnr:.1f} dB", flush=True)
