"""Builds and runs the C++ host-side mirror test (tests/cpp/test_core_api.cpp) against
libislands_amd.so: the CPU half here, the device half under -m gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "islands_amd", "lib")
EXE = os.path.join(LIBDIR, "test_core_api")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_core_api.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src,
                           "-L", LIBDIR, "-lislands_amd", f"-Wl,-rpath,{LIBDIR}", "-o", EXE])


def _run(mode):
    return subprocess.run([EXE, mode], capture_output=True, text=True, timeout=300)


def test_cpp_mirror_cpu():
    _build()
    r = _run("cpu")
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_mirror_gpu():
    _build()
    r = _run("gpu")
    assert r.returncode == 0, r.stdout + r.stderr
