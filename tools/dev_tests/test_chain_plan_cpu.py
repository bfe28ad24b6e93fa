"""CPU: the dependency planning of the cross-layer chain (csrc/chain_plan.h - which K-chunks of a layer wait for which counter
value) compiled with g++ and checked on the RDB layout Model::forward produces."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_chain_plan_harness(tmp_path):
    exe = str(tmp_path / "chain_plan_harness")
    src = os.path.join(ROOT, "tools", "dev_tests", "chain_plan_harness.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "chain plan ok" in r.stdout, r.stdout + r.stderr
