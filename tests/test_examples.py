"""The C++ mirror classes (eagle-mpc_amd/host/eagle_mpc.hpp) and the C ABI itself end to end through the example programs.

CPU: the examples build against libempc.so and fail loudly without a GPU (no CPU fallback anywhere).
GPU: they reproduce the oracle's solution of the displacement problem and run the closed loop.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


@pytest.fixture(scope="module")
def built(empc):
    subprocess.check_call(["make", "-C", EX, "-s"])
    return {n: os.path.join(EX, "cpp", n) for n in ("trajectory", "mpc")}


def run(path):
    return subprocess.run([path, ROOT], capture_output=True, text=True, timeout=600)


def test_examples_build_and_refuse_to_run_without_gpu(built, empc):
    if empc.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    for p in built.values():
        r = run(p)
        assert r.returncode == 1
        assert "no HIP device available" in r.stderr and "no CPU fallback" in r.stderr


def test_plain_c_example_of_the_c_abi(built, empc):
    """examples/c/trajectory.c: the boundary used from C99 (gcc -std=c99 -pedantic: include/empc.h is a C header, not only a C++
    one).  The host side -- YAML factory, shooting problem, the solver query -- needs no GPU; the solve does."""
    exe = os.path.join(EX, "c", "trajectory")
    r = subprocess.run([exe, ROOT, "hexacopter370_flying_arm_3/trajectories/eagle_catch.yaml", "32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "nx 19 ndx 18 nu 9" in r.stdout and "contact dynamics: yes" in r.stdout and "stage grasp" in r.stdout
    assert "T = 99 knots" in r.stdout and "a kernel instantiation exists" in r.stdout
    if empc.device_count() > 0:
        m = re.search(r"iterations (\d+) cost ([0-9.]+)", r.stdout)
        assert m and int(m.group(1)) == 64 and "kernels: baked hexacopter370_flying_arm_3, ContactModel3D" in r.stdout
    else:
        assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_examples_on_gpu(built):
    r = run(built["trajectory"])
    assert r.returncode == 0, r.stderr
    m = re.search(r"iterations (\d+) cost ([0-9.]+)", r.stdout)
    assert m and int(m.group(1)) == 17 and abs(float(m.group(2)) - 124.571017) < 1e-5   # tests/golden/solutions/displacement.npz
    r = run(built["mpc"])
    assert r.returncode == 0, r.stderr
    m = re.search(r"t = 200 ms: plant position ([-0-9.e]+) ([-0-9.e]+) ([-0-9.e]+)", r.stdout)
    assert m
    z = float(m.group(3))
    assert 0.0 < z < 0.2   # the plant climbs along the planned trajectory (reference z at 200 ms ~ 0.05 m)


@pytest.mark.gpu
def test_python_examples_on_gpu():
    import sys
    for name, pattern in (("trajectory.py", r"iterations 17 cost 124\.5710"), ("mpc.py", r"tracking error \(position, max over plants\): 0\.\d+ m")):
        r = subprocess.run([sys.executable, os.path.join(EX, "python", name)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        assert re.search(pattern, r.stdout), r.stdout
