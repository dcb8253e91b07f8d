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
    assert "OK" in out.stdout


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
    assert "OK" in out.stdout


PARSAC_SRC = os.path.join(ROOT, "tests", "cpp", "parsac_device_test.cpp")
PARSAC_EXE = os.path.join(ROOT, "tests", "cpp", "parsac_device_test.bin")


def _compile_parsac():
    rbuild.build()
    libdir = os.path.join(ROOT, "rd_vio_amd")
    # -ffp-contract=off like librdvio_pipeline.so: the host scoring must round exactly like the device kernel
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-ffp-contract=off", "-o", PARSAC_EXE, PARSAC_SRC, "-L", libdir, "-lrdvio_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])


def test_parsac_device_test_compiles_and_links():
    _compile_parsac()
    assert os.path.exists(PARSAC_EXE)


@pytest.mark.gpu
def test_parsac_device_scoring_matches_host_scoring():
    """row N2: rdvio_hip_parsac_score / _fetch against the host road of parsac.hpp -- masks, models, bin confidences bit-identical"""
    _compile_parsac()
    out = subprocess.run([PARSAC_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout


GATES_SRC = os.path.join(ROOT, "tests", "cpp", "gates_device_test.cpp")
GATES_EXE = os.path.join(ROOT, "tests", "cpp", "gates_device_test.bin")


def _compile_gates():
    rbuild.build()
    libdir = os.path.join(ROOT, "rd_vio_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-ffp-contract=off", "-o", GATES_EXE, GATES_SRC, "-L", libdir, "-lrdvio_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])


def test_gates_device_test_compiles_and_links():
    _compile_gates()
    assert os.path.exists(GATES_EXE)


@pytest.mark.gpu
def test_two_view_gates_and_thinning_on_the_device_match_the_host_road():
    """row N3: rdvio_hip_ransac_generate_score / _ransac_fetch / rdvio_hip_thin_tracks against geom.hpp's host road -- essential and
    rotation gate models and masks, keep flags of the track-length thinning: bit-identical"""
    _compile_gates()
    out = subprocess.run([GATES_EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK" in out.stdout
