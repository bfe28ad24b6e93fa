"""GPU: the stream-pair test (``ss4k_stream_pair_check``) and the context's self-check of its lane stream.

HIP maps a process's streams onto a few hardware queues; two streams of one queue run in order, and the 5th queue of a process is slow
against the NULL stream's while both are busy (``profiles/earlier/r05/r05_lane_queue.txt``: a 4-frame RRDBNet job with two launch chains 110 instead of
117 frames/s when three streams had been created before the context's lane stream).  ``ss4k_ctx::lane_check`` (csrc/models.cpp) measures the
pair before the first fork and replaces a lane stream that fails.
"""
import os
import subprocess
import sys

import pytest
import torch

import sharkshark4k_amd  # noqa: F401
from sharkshark4k_amd import _capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_pair_check_answers_and_rejects_bad_arguments(ctx):
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    assert ctx.streams_side_by_side(a, b) in (True, False)
    assert ctx.streams_side_by_side(torch.cuda.current_stream(), a) in (True, False)
    with pytest.raises(_capi.Ss4kError, match="same stream"):
        ctx.streams_side_by_side(a, a)
    rc = _capi.lib().ss4k_stream_pair_check(ctx._h, int(a.cuda_stream), int(b.cuda_stream), None)
    assert rc == -22
    # the streams are usable afterwards and ordinary work on them is untouched
    with torch.cuda.stream(a):
        x = torch.arange(1000, device="cuda").sum()
    a.synchronize()
    assert int(x) == 499500


SCRIPT = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
import sharkshark4k_amd
from sharkshark4k_amd import _capi, weights as W
from tests.helpers import smooth_u8
torch.zeros(1, device="cuda")                              # the NULL stream takes the process's first hardware queue ...
keep = [torch.cuda.Stream() for _ in range({k})]          # ... K used streams the next K: the context's lane stream gets the (K+2)th
for st in keep:
    with torch.cuda.stream(st): torch.zeros(1, device="cuda")
torch.cuda.synchronize()
flat = W.flatten(W.rrdbnet_table(5, scale=2, num_block=2), W.rrdbnet_keys(2))
x = torch.from_numpy(smooth_u8(9, (4, 96, 136, 3))).permute(0, 3, 1, 2).float().div(255.0).cuda()
outs = []
for flags in (_capi.MODEL_TWO_CHAINS, _capi.MODEL_ONE_CHAIN):
    ctx = _capi.Context(0)
    m = _capi.Model(ctx, _capi.make_desc(_capi.RRDBNET, _capi.F16, scale=2, num_block=2, flags=flags), flat)
    outs.append(m(x).cpu()); outs.append(m(x).cpu())
assert all(torch.equal(outs[0], o) for o in outs[1:]), "two chains != one chain"
print("BIT IDENTICAL")
"""


@pytest.mark.parametrize("k", [0, 3])
def test_lane_stream_is_checked_once_and_a_bad_one_replaced(k):
    """K = 3 with eight hardware queues is the pairing that was slow in round 5 (the 1st and the 5th queue a process takes; 5 boxes of 5): if it
    shows, the log has a replacement and then a stream that passes; everywhere the check runs ONCE per (context, caller stream) - the second forward adds no line - a one-chain model never
    runs it, and the results equal the one-chain model's bit for bit whichever stream serves the second chain."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", SS4K_LANE_CHECK_LOG="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, k=k)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BIT IDENTICAL" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[streams]")]
    print("\n".join(lines))
    assert 1 <= len(lines) <= 6
    assert all("NOT side by side" in ln for ln in lines[:-1])          # every line but the last is a rejected stream ...
    assert lines[-1].endswith("-> side by side")                       # ... and the one in use passed
