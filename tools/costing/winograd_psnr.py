"""Costing only (VERDICT r2 "next" 8b; nothing here ships): what would Winograd F(2x2, 3x3) do to the accuracy of the fp16 RRDBNet path?

Emulates, in PyTorch on the CPU, the 23-block RRDBNet x2 with every 3x3 convolution computed
  (a) directly, fp16 storage of activations and weights, fp32 accumulation   (what csrc/conv_mfma.hip / conv_rs.hip do), and
  (b) as F(2x2, 3x3): U = G g G^T and V = B^T d B rounded to fp16 (the MFMA operands), the 16 element-wise products accumulated
      over the input channels in fp32, Y = A^T M A in fp32, one fp16 rounding of the stored activation,
and reports the PSNR of both against the fp32 network on the same weights and input.  2.25x fewer multiplies per output; the
transformed input grows by up to 4x in magnitude (B^T d B sums four taps with signs), the transformed weights carry factors 1/2 and
1/4 - both lose fp16 bits before the contraction.

usage: python tools/costing/winograd_psnr.py [height=96] [width=160] [blocks=23]
"""
import math, sys, time
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, __file__.rsplit("/", 3)[0])
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import weights as W

H = int(sys.argv[1]) if len(sys.argv) > 1 else 96
Wd = int(sys.argv[2]) if len(sys.argv) > 2 else 160
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 23
torch.set_num_threads(8)

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
h16 = lambda t: t.half().float()


def conv_direct(x, w, b, mode):
    if mode == "f32":
        return F.conv2d(x, w, b, padding=1)
    return F.conv2d(h16(x), h16(w), b, padding=1)           # fp16 operands, fp32 accumulation


def conv_winograd(x, w, b):
    n, c, h, ww = x.shape
    hp, wp = (h + 1) // 2 * 2, (ww + 1) // 2 * 2
    xp = F.pad(h16(x), (1, 1 + wp - ww, 1, 1 + hp - h))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                    # (n, c, th, tw, 4, 4)
    V = h16(torch.einsum("ij,nctujk,lk->nctuil", BT, d, BT))  # B^T d B, rounded to the MFMA operand type
    U = h16(torch.einsum("ij,ocjk,lk->ocil", G, h16(w), G))   # G g G^T
    M = torch.einsum("ocil,nctuil->notuil", U, V)             # fp32 accumulation over the input channels
    Y = torch.einsum("ij,notujk,lk->notuil", AT, M, AT)       # (n, o, th, tw, 2, 2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], hp, wp)[:, :, :h, :ww]
    return y + b.view(1, -1, 1, 1)


def rrdbnet(x, t, conv, store):
    lr = lambda v: F.leaky_relu(v, 0.2)
    g = lambda k: torch.from_numpy(np.asarray(t[k]))
    cv = lambda v, name: conv(v, g(name + ".weight"), g(name + ".bias"))
    feat = store(cv(F.pixel_unshuffle(x, 2), "conv_first"))
    body = feat
    for i in range(NB):
        blk = body
        for r in (1, 2, 3):
            p = f"body.{i}.rdb{r}"
            xs = [blk]
            for c in range(1, 5):
                xs.append(store(lr(cv(torch.cat(xs, 1), f"{p}.conv{c}"))))
            blk = store(cv(torch.cat(xs, 1), f"{p}.conv5") * 0.2 + blk)
        body = store(blk * 0.2 + body)
    feat = store(feat + cv(body, "conv_body"))
    feat = store(lr(cv(F.interpolate(feat, scale_factor=2, mode="nearest"), "conv_up1")))
    feat = store(lr(cv(F.interpolate(feat, scale_factor=2, mode="nearest"), "conv_up2")))
    return cv(store(lr(cv(feat, "conv_hr"))), "conv_last")


def psnr(a, b, peak):
    return 10 * math.log10(peak * peak / float(((a - b) ** 2).mean()))


t = dict(W.rrdbnet_table(0, scale=2, num_block=NB))
t["conv_last.weight"] = t["conv_last.weight"] * np.float32(0.01)      # image-range output, as in the parity tests
t["conv_last.bias"] = np.full_like(t["conv_last.bias"], 0.5)
g = torch.Generator().manual_seed(0)
x = F.avg_pool2d(torch.rand(1, 3, H + 8, Wd + 8, generator=g), 9, 1)     # smooth image in [0, 1]
with torch.no_grad():
    t0 = time.time()
    ref = rrdbnet(x, t, lambda v, w, b: conv_direct(v, w, b, "f32"), lambda v: v)
    d16 = rrdbnet(x, t, lambda v, w, b: conv_direct(v, w, b, "f16"), h16)
    wg = rrdbnet(x, t, conv_winograd, h16)
print(f"RRDBNet x2, {NB} blocks, {H}x{Wd} input, output range [{float(ref.min()):.3f}, {float(ref.max()):.3f}]  ({time.time() - t0:.0f} s)")
print(f"  direct conv, fp16 operands / fp32 accumulate vs fp32 network : PSNR {psnr(d16, ref, 1.0):6.2f} dB, max |err| {float((d16 - ref).abs().max()):.2e}")
print(f"  Winograd F(2x2,3x3), fp16 U and V / fp32 accumulate         : PSNR {psnr(wg, ref, 1.0):6.2f} dB, max |err| {float((wg - ref).abs().max()):.2e}")
print(f"  8-bit frames: 1 LSB = {1 / 255:.2e}")
