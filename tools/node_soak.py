#!/usr/bin/env python3
"""Soak of the multi-worker node with workers dying and being replaced (GPU box; two workers on cuda:0).

For `seconds`: host frames through ``UpscalerNode(devices=[0, 0])`` (the product path: pinned rings, in-worker copy streams, ordered fan-in),
every result compared byte for byte with an in-process upscaler; every `kill_every` seconds one worker (alternating) is killed with
SIGKILL in mid-stream, the stream goes on over the survivor (the host steps that were inside the victim are re-queued from its input
ring: ``report()['rescued']``), ``replace_dead()`` starts a fresh child in the slot.  Checked: zero wrong frames; every step either arrives
in order or is counted lost, and only steps that were inside the killed worker can be lost; the
SURVIVOR's device memory does not grow (``hipMemGetInfo`` before / after, this process holds the reference model only).

usage: python tools/node_soak.py [seconds=300] [kill_every=30]
"""
import os
import signal
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import sharkshark4k_amd  # noqa: E402,F401
from sharkshark4k_amd import _capi, weights as W  # noqa: E402
from sharkshark4k_amd.node import UpscalerNode  # noqa: E402
from tests.helpers import smooth_u8  # noqa: E402

if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    kill_every = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    LR = (180, 320)
    KW = dict(upscaler_model="realesrgan", model_name="RealESRGAN_x2plus", denoising=False, weights="synthetic", seed=3, lr_shape=LR, dtype="f16")
    ctx = _capi.Context(0)
    sr = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2), W.flatten(W.rrdbnet_table(3, scale=2), W.rrdbnet_keys(23)))
    up = _capi.Upscaler(ctx, sr, LR, None, True, False, None, 1.0)
    pool = smooth_u8(5, (16, LR[0], LR[1], 3))
    ref = torch.cat([up(torch.from_numpy(pool[i:i + 1]).cuda()) for i in range(16)]).cpu()
    torch.cuda.synchronize()
    node = UpscalerNode(devices=[0, 0], fps=24, frame_skips=False, lost_after_s=3.0, host_slots=6, **KW)
    node.start(timeout=600)
    src_of = {}          # step -> first frame index of its four frames
    rng = np.random.default_rng(0)
    good = bad = emitted = kills = 0
    inside_killed = 0
    free_marks = []

    def free_mb():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0] / 2 ** 20

    def pump(until, max_in_flight=6):
        global good, bad, emitted
        while time.monotonic() < until:
            if node.dispatcher.frame_step - node.dispatcher.next_emit < max_in_flight:
                a = int(rng.integers(0, 13))
                for s in node.submit_batch(pool[a:a + 4]):
                    src_of[s] = a
            for e in node.poll(0.002):
                a = src_of.pop(e.step)
                emitted += 1
                if torch.equal(e.frames, ref[a:a + 4]):
                    good += 1
                else:
                    bad += 1
                    print(f"step {e.step}: MISMATCH", flush=True)

    t_end = time.monotonic() + seconds
    try:
        pump(time.monotonic() + 5.0)
        free_marks.append(free_mb())
        victim = 1
        while time.monotonic() < t_end:
            pump(min(t_end, time.monotonic() + kill_every))
            if time.monotonic() >= t_end:
                break
            svc = node.services[victim]
            lost_before = node.report()["lost"]
            inside = sum(1 for s, o in node.dispatcher._owner.items() if o is svc)     # steps queued in / running on the victim right now
            os.kill(svc.proc.pid, signal.SIGKILL)                                      # exactly the child this node started
            svc.proc.join(30)
            kills += 1
            pump(time.monotonic() + 4.0)                                               # the stream goes on over the survivor
            lost_now = node.report()["lost"] - lost_before
            inside_killed += inside
            print(f"[{seconds - (t_end - time.monotonic()):6.1f} s] killed worker {victim} with {inside} step(s) inside: {lost_now} lost, alive {node.alive()}, "
                  f"{good} good / {bad} bad so far, free {free_mb():.0f} MB", flush=True)
            assert lost_now <= inside, f"{lost_now} steps lost but only {inside} were inside the killed worker"
            node.replace_dead(timeout=600)
            pump(time.monotonic() + 3.0)
            free_marks.append(free_mb())                                               # two workers up again, steady state
            victim ^= 1
        pump(time.monotonic() + 1.0, max_in_flight=0)
        rep = node.report()
    finally:
        node.stop()
        node.close()
    growth = free_marks[0] - min(free_marks[1:]) if len(free_marks) > 1 else 0.0
    print(f"{seconds:.0f} s, {kills} kill/replace cycles: {good} jobs right, {bad} wrong, {rep['lost']} lost, {rep['rescued']} rescued ({inside_killed} were inside killed workers), "
          f"{rep['rerouted']} rerouted, free device memory with both workers up: {free_marks[0]:.0f} MB at the start, min {min(free_marks):.0f} MB later "
          f"(growth {growth:.0f} MB)")
    ok = bad == 0 and rep["lost"] <= inside_killed and growth < 256 and good > 0
    print("NODE SOAK OK" if ok else "NODE SOAK FAILED")
    sys.exit(0 if ok else 1)
