"""GPU: kernels that exist in the dev library only (libss4k_hip_dev.so: measured experiments that did not become a product route) keep
working - their own tests (tools/dev_tests/) run in a child process that binds the dev library.

conv_d16.hip: the fused dense-block layer pairs on v_mfma_f32_16x16x32_f16 with 14 x 32 tiles (SS4K_D16=1).  Correct on the first run and
as accurate as the 32x32x16 build, level with it at two and four frames per job and 10 % behind on one 720p frame (520 tiles of 14 rows
for 512 workgroup slots) - profiles/NOTES_r04.md."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dev_library_d16_suite():
    from sharkshark4k_amd import build as B
    assert os.path.exists(B.LIB_DEV), "libss4k_hip_dev.so was not built (__graft_entry__.build())"
    env = dict(os.environ, SS4K_LIB=B.LIB_DEV, SS4K_D16="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tools", "dev_tests", "test_d16.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout


def _run_dev_suite(files, extra_env=None, timeout=1500):
    from sharkshark4k_amd import build as B
    assert os.path.exists(B.LIB_DEV), "libss4k_hip_dev.so was not built (__graft_entry__.build())"
    env = dict(os.environ, SS4K_LIB=B.LIB_DEV, **(extra_env or {}))
    r = subprocess.run([sys.executable, "-m", "pytest", *[os.path.join(ROOT, "tools", "dev_tests", f) for f in files], "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1500:]
    assert " passed" in r.stdout


def test_dev_library_conv_rs_suite():
    _run_dev_suite(["test_conv_rs.py", "test_wide_rrdbnet.py"])


def test_dev_library_chain_suite():
    _run_dev_suite(["test_chain.py", "test_chain_plan_cpu.py"])
