"""Dev tool: run BSVD-32 720p n=4 a few times (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd
from sharkshark4k_amd import _capi
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
bs = factory.build_denoise_model(ctx, weights="synthetic", dtype="f16")
x = torch.rand(int(sys.argv[1]) if len(sys.argv) > 1 else 4, 4, 720, 1280, device="cuda")
for _ in range(4): bs(x)
torch.cuda.synchronize()
