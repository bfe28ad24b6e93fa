"""Dev tool: time the RRDBNet x2 tail layer shapes at 4 frames (conv_last is the GEN/NCHW build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
m = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", dtype="f16", scale=2, num_block=1, weights=W.rrdbnet_table(3, scale=2, num_block=1))
x = torch.rand(4, 3, 720, 1280, device="cuda")
for _ in range(3): m(x)
torch.cuda.synchronize()
