"""The C++ host mirror (rd_vio_amd/host/rdvio_hip.hpp): compiles against the C ABI without a GPU; on the GPU box
the driver in tests/cpp/ runs the reference-style call sequence (detect -> track -> preintegrate -> solve)."""
import os
import subprocess

import pytest

from rd_vio_amd import build as rbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.bin")


def _compile():
    rbuild.build()
    libdir = os.path.join(ROOT, "rd_vio_amd")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, SRC, "-L", libdir, "-lrdvio_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_host_mirror_compiles_and_links():
    _compile()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_mirror_runs_on_gpu():
    _compile()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")


ODO_SRC = os.path.join(ROOT, "tests", "cpp", "odometry_mirror_test.cpp")
ODO_EXE = os.path.join(ROOT, "tests", "cpp", "odometry_mirror_test.bin")


def _compile_odometry():
    rbuild.build()
    libdir = os.path.join(ROOT, "rd_vio_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", ODO_EXE, ODO_SRC, "-L", libdir, "-lrdvio_pipeline", "-lrdvio_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])


def test_odometry_mirror_compiles_and_links():
    _compile_odometry()
    assert os.path.exists(ODO_EXE)


@pytest.mark.gpu
def test_odometry_mirror_runs_on_gpu():
    _compile_odometry()
    out = subprocess.run([ODO_EXE], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK")
