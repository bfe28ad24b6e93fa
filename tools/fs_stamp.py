"""Dev tool (GPU box, dev library): phase cycle stamps of the fp16-mode FSRCNN mapping stage, per layer wave.
usage: SS4K_LIB=$PWD/sharkshark-4k_amd/libss4k_hip_dev.so SS4K_FS_STAMP=1 [SS4K_MH_NU=2] python3 tools/fs_stamp.py   (prints on stderr)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
ctx = _capi.Context(0)
x = torch.rand(12, 1, 720, 1280, generator=torch.Generator().manual_seed(5)).cuda()
m = factory.build_model_fsrcnn(ctx, factor=2, weights=W.fsrcnn_table(seed=2), dtype="f16")
for _ in range(3): m(x)
torch.cuda.synchronize()
