"""GPU: lifetimes.  Device memory comes back after models / upscalers are destroyed and rebuilt, an image-mode service fed arbitrary shapes
stays bounded, a node survives kill / replace cycles (tools/node_soak.py).  ``hipMemGetInfo`` (``torch.cuda.mem_get_info``) is the witness:
the library allocates with hipMalloc directly, torch's own cache is emptied before every reading."""
import gc
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi
from sharkshark4k_amd import weights as W
from sharkshark4k_amd.upscale import model as factory
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MB = 2 ** 20


def free_mb():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(0)[0] / MB


def test_model_create_destroy_cycles_give_the_memory_back(ctx):
    """50 x create / forward / destroy of each network family: free device memory returns to where it started (a leak of one weight blob
    per cycle would be 33 MB x 50 for RRDBNet; of one workspace, GBs)."""
    x = torch.rand(1, 3, 96, 128, device="cuda")
    def family(desc, flat):     # (the weights are generated once: the cycle is create -> forward -> destroy of the DEVICE object)
        return lambda: _capi.Model(ctx, desc, flat)
    builds = {
        "rrdbnet23_f16": family(factory.esrgan_desc("RealESRGAN_x2plus", "f16"), factory.esrgan_flat("RealESRGAN_x2plus", 0.5, "synthetic", 0)),
        "srvgg32_f16": family(factory.esrgan_desc("realesr-general-x4v3", "f16"), factory.esrgan_flat("realesr-general-x4v3", 0.5, "synthetic", 0)),
        "bsvd32_f16": family(factory.denoise_desc("f16"), factory.denoise_flat("synthetic", 0)),
    }
    for name, build in builds.items():
        m = build()
        xin = torch.rand(1, 1, 4, 96, 128, device="cuda") if name.startswith("bsvd") else x
        m(xin); del m
        start = free_mb()
        lows = []
        for i in range(50):
            m = build()
            m(xin)
            torch.cuda.synchronize()
            if i % 10 == 0:
                lows.append(torch.cuda.mem_get_info(0)[0] / MB)
            m.close()
            del m
        end = free_mb()
        print(f"{name}: free {start:.0f} MB before, {end:.0f} MB after 50 cycles (while alive: {min(lows):.0f} MB)")
        assert end > start - 8, f"{name}: {start - end:.0f} MB did not come back after 50 create / destroy cycles"
        assert min(lows) < start - 1, "the models did allocate something (the witness works)"


def test_upscaler_rebuilds_with_changing_output_shape(ctx):
    """20 x a new upscaler on the same model with another ``output_shape`` (what a service does when the pipeline overwrites the
    attribute): scratch buffers of the old one are released."""
    table = W.rrdbnet_table(1, scale=2, num_block=2)
    sr = factory.build_model_esrgan(ctx, "RealESRGAN_x2plus", weights=table, dtype="f16", num_block=2)
    frames = torch.randint(0, 256, (2, 180, 320, 3), dtype=torch.uint8, device="cuda")
    up = _capi.Upscaler(ctx, sr, (180, 320), None, True, False, None, 1.0)
    up(frames); up.close(); del up
    start = free_mb()
    for i in range(20):
        shape = (360 + 24 * (i % 5), 640 + 32 * (i % 7))
        up = _capi.Upscaler(ctx, sr, (180, 320), shape, True, False, None, 1.0)
        out = up(frames)
        assert out.shape == (2, shape[0], shape[1], 3)
        torch.cuda.synchronize()
        up.close()
        del up, out
    end = free_mb()
    print(f"upscaler rebuilds: free {start:.0f} MB before, {end:.0f} MB after")
    assert end > start - 8, f"{start - end:.0f} MB did not come back after 20 upscaler rebuilds"
    # ... and through the service object, the way a pipeline does it: the attribute changes, the next job gets a new upscaler
    svc = HipUpscalerService(device=0, upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=1,
                             lr_shape=(180, 320), dtype="f16", overlap_jobs=False)
    svc.proc_init()
    svc.upscale(frames)
    start = free_mb()
    for i in range(20):
        svc.output_shape = (360 + 24 * (i % 5), 640 + 32 * (i % 7))
        assert svc.upscale(frames).shape[1:3] == svc.output_shape
    torch.cuda.synchronize()
    end = free_mb()
    print(f"service output_shape changes: free {start:.0f} MB before, {end:.0f} MB after")
    assert end > start - 64      # (the largest shape's buffers stay with the live upscaler)


def test_image_mode_random_shapes_memory_bounded(ctx):
    """The image server's caller: one-frame jobs of ARBITRARY shape (up to 4096 x 2048).  200 random shapes through one service (three job
    sets): device memory stays bounded by the largest shape's needs - it does not grow with the number of distinct shapes - and the
    per-shape caches are bounded."""
    svc = HipUpscalerService(lr_level=3, device=0, denoising=False, denoise_rate=0.2, upscaler_model="realesrgan", batch_size=1, jit_mode=False,
                             lr_hr_resize=False, model_name="RealESRGAN_x2plus", weights="synthetic", seed=0, dtype="f16")
    svc.proc_init()
    rng = np.random.default_rng(0)
    big = torch.randint(0, 256, (1, 1024, 2048, 3), dtype=torch.uint8, device="cuda")   # the largest shape of this run first: it sets the bound
    for _ in range(3):
        svc.upscale(big)
    after_big = free_mb()
    lows = []
    for i in range(200):
        h, w = 2 * int(rng.integers(8, 512)), 2 * int(rng.integers(36, 1024))     # (x2 RRDBNet un-shuffles by 2: even sizes, as the reference's network needs)
        f = torch.randint(0, 256, (1, h, w, 3), dtype=torch.uint8, device="cuda")
        out = svc.upscale(f)
        assert out.shape == (1, 2 * h, 2 * w, 3)
        if i % 20 == 19:
            lows.append(free_mb())
    print(f"image mode: free {after_big:.0f} MB after the largest shape, {min(lows):.0f} .. {max(lows):.0f} MB over 200 random shapes; "
          f"{len(svc._small)} shapes in the small-job cache")
    assert min(lows) > after_big - 512, f"device memory kept shrinking with new shapes: {after_big - min(lows):.0f} MB below the largest shape's level"
    assert len(svc._small) <= 257
    # the maximum the server admits (4096 x 2048) still runs afterwards
    out = svc.upscale(torch.randint(0, 256, (1, 2048, 4096, 3), dtype=torch.uint8, device="cuda"))
    assert out.shape == (1, 4096, 8192, 3)
    torch.cuda.synchronize()


def test_node_soak_with_kill_and_replace_cycles():
    """tools/node_soak.py for 75 s with a kill + replace_dead() every 20 s (the 5-minute run with 30 s cycles is recorded in
    profiles/r06_node_soak.txt): no wrong frame, only steps inside a killed worker are lost, the survivor's memory does not grow."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "node_soak.py"), "75", "20"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0 and "NODE SOAK OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
