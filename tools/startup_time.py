import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0=time.time()
import torch, sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale import model as factory
t1=time.time()
ctx=_capi.Context(0)
tab=W.rrdbnet_table(0, scale=2); t2=time.time()
m=factory.build_model_esrgan(ctx,"RealESRGAN_x2plus",weights=tab,dtype="f16",scale=2); torch.cuda.synchronize(); t3=time.time()
x=torch.rand(1,3,720,1280,device="cuda"); y=m(x); torch.cuda.synchronize(); t4=time.time()
print(f"import {t1-t0:.2f}s  synthetic weight table {t2-t1:.2f}s  model create (pack+upload) {t3-t2:.2f}s  first forward {t4-t3:.3f}s")
