"""Build libss4k_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

``build(dev=True)`` builds libss4k_hip_dev.so from the same sources with -DSS4K_DEV: the product ABI plus
``ss4k_bench_conv`` (include/ss4k_dev.h) and the instrumented / alternative-shape instantiations of the
conv kernel that the measurement tools use.  The product library carries none of that."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libss4k_hip.so")
LIB_DEV = os.path.join(HERE, "libss4k_hip_dev.so")
SOURCES = ["conv_mfma.hip", "conv_pair.hip", "conv_dense.hip", "conv_w16.hip", "conv_w16n.hip", "glue.hip", "fsrcnn.hip", "pack.cpp", "models.cpp", "api.cpp"]
# libss4k_hip_dev.so alone: measured experiments (conv_s3, conv_d16) and the kernels of rounds 2-4 that no product route uses any more
# (conv_rs: register-stationary weights; conv_chain: the RRDB body as one persistent launch) - tools/dev_tests/ keeps them honest
DEV_SOURCES = ["conv_s3.hip", "conv_d16.hip", "conv_rs.hip", "conv_chain.hip"]
# conv_rs.hip: the per-tile body is thousands of fully unrolled MFMAs (weights live in named registers); hipcc's
# default cap on '#pragma unroll' size would leave the chunk loop rolled and the weights in scratch
# fsrcnn.hip: no SLP vectorisation - left on, the tail's overlap-add (two adjacent output columns per lane) is packed into v_pk_add_f32,
# which cannot take a DPP operand: 20 of its 35 wave shifts per row then become separate v_mov_b32_dpp, and packed fp32 adds are slower
# than two plain ones beside MFMAs (MI355X_MICROARCH.md, cycle constants).  Without it every shift is folded into its add (v_add_f32_dpp)
EXTRA_FLAGS = {"conv_rs.hip": ["-mllvm", "-pragma-unroll-threshold=4000000"], "fsrcnn.hip": ["-fno-slp-vectorize"]}
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _needs_build(obj: str, src: str) -> bool:
    if not os.path.exists(obj):
        return True
    newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))
    newest = max(newest, os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "..", "include", "ss4k.h")),
                 os.path.getmtime(os.path.join(HERE, "..", "include", "ss4k_dev.h")))
    return os.path.getmtime(obj) < newest


def build(force: bool = False, verbose: bool = True, dev: bool = False) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build_dev" if dev else "build")
    lib = LIB_DEV if dev else LIB
    flags = FLAGS + (["-DSS4K_DEV"] if dev else [])
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    sources = SOURCES + (DEV_SOURCES if dev else [])
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        if force or _needs_build(obj, src):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in sources]
    if jobs or not os.path.exists(lib):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv))
