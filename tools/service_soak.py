"""Dev tool (GPU box): soak of the service worker path with the one-frame overlap - N jobs of mixed sizes (1, 1, 1, 2, 1, 4, 1, ... frames, fresh
input tensors that the producer drops right after pushing) through ONE spawned HipUpscalerService worker, every result compared byte for byte with
an in-process single-set upscaler.  Catches stream-ordering / tensor-lifetime races of the job sets (hip_upscaler.py).
usage: python3 tools/service_soak.py [jobs=600]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sharkshark4k_amd  # noqa
from sharkshark4k_amd import _capi, weights as W
from sharkshark4k_amd.upscale.hip_upscaler import HipUpscalerService
from sharkshark4k_amd.upscale.upscaler_base import UpscalerQueueEntry
from tests.helpers import smooth_u8

class NoHold(HipUpscalerService):
    """--no-hold: the service WITHOUT its hold on a job's input (what round 5's first build did): shows that the soak sees the race."""
    def _retire(self):
        self._inflight.clear()


if __name__ == "__main__":
    no_hold = "--no-hold" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    njobs = int(args[0]) if args else 600
    LR = (180, 320)
    KW = dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=3, lr_shape=LR, dtype="f16")
    ctx = _capi.Context(0)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(W.rrdbnet_table(3, scale=2), W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, LR, None, True, False, None, 1.0)
    pool = torch.from_numpy(smooth_u8(5, (16, LR[0], LR[1], 3))).cuda()
    ref = torch.cat([up(pool[i:i + 1]) for i in range(16)]).cpu()
    svc = (NoHold if no_hold else HipUpscalerService)(device=0, **KW)
    svc.start()
    rng = np.random.default_rng(0)
    sizes = [int(rng.choice([1, 1, 1, 1, 2, 4])) for _ in range(njobs)]
    starts = [int(rng.integers(0, 16 - n + 1)) for n in sizes]
    sent = got = bad = 0
    t0 = time.perf_counter()
    try:
        while got < njobs:
            while sent < njobs and sent - got < 10:
                x = pool[starts[sent]:starts[sent] + sizes[sent]].clone()   # a fresh tensor per job, dropped by the producer right after the push
                svc.push_job(UpscalerQueueEntry(frames=x, step=sent), timeout=600)
                del x
                sent += 1
            e = svc.get_result(timeout=600)
            assert e.step == got, (e.step, got)
            want = ref[starts[got]:starts[got] + sizes[got]]
            if not torch.equal(e.frames.cpu(), want):
                bad += 1
                print(f"job {got} ({sizes[got]} frames from {starts[got]}): MISMATCH in {int((e.frames.cpu() != want).sum())} bytes", flush=True)
            got += 1
    finally:
        svc.stop()
    dt = time.perf_counter() - t0
    print(f"{njobs} jobs ({sum(sizes)} frames) through one worker in {dt:.1f} s, {bad} mismatching:", "SERVICE SOAK FAILED" if bad else "SERVICE SOAK OK")
    sys.exit(1 if bad else 0)
