"""Builds librdvio_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdvio_hip.so")
SOURCES = ["capi.hip", "image_kernels.hip", "lk_kernels.hip", "ba_kernels.hip", "select_kernels.hip", "parsac_kernels.hip", "solver_kernels.hip", "solver_host.hip", "marg_host.hip",
           "host_select.cpp", "sequences.cpp"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# No FMA contraction anywhere: the image / LK arithmetic must round exactly like the oracle (bit-exact feature
# indices), and the FP64 estimation code relies on exact cancellations the reference (built without FMA) also has --
# e.g. q0^-1 * q == identity at the linearisation point, which the 1e15 prior pin would otherwise amplify to 1e-2.
CONTRACT = {}


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "rdvio_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


PIPE_DIR = os.path.join(HERE, "host", "pipeline")
PIPE_LIB = os.path.join(HERE, "librdvio_pipeline.so")
PIPE_SOURCES = ["map.cpp", "pipeline.cpp", "initializer.cpp", "hip_backend.cpp"]


def build_pipeline(force=False, verbose=False):
    """librdvio_pipeline.so: the host orchestration (plain C++17, no device code) on top of librdvio_hip.so."""
    deps = [os.path.join(PIPE_DIR, f) for f in os.listdir(PIPE_DIR)] + [os.path.join(HERE, "..", "include", f)
                                                                          for f in ("rdvio_hip.h", "rdvio_pipeline.h")] + [LIB]
    if not force and os.path.exists(PIPE_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(PIPE_LIB) for d in deps):
        return PIPE_LIB
    cxx = os.environ.get("CXX", "g++")
    # same no-FMA-contraction rule as the device code: the CPU-path comparison runs this very code over the oracle
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-ffp-contract=off", "-pthread", "-o", PIPE_LIB] + \
          [os.path.join(PIPE_DIR, s) for s in PIPE_SOURCES] + ["-L" + HERE, "-lrdvio_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return PIPE_LIB


def build_pipeline_variant(name, flags):
    """librdvio_pipeline.so compiled with other optimisation flags (bench.py's CPU-baseline builds: the orchestration is part
    of the CPU path that is timed) into rd_vio_amd/_build/; -ffp-contract=off stays, so results do not change."""
    import hashlib

    try:
        key = "".join(l for l in open("/proc/cpuinfo").read().splitlines() if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        key = "unknown"
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"librdvio_pipeline_{name}_{hashlib.sha1(key.encode()).hexdigest()[:10]}.so")
    deps = [os.path.join(PIPE_DIR, f) for f in os.listdir(PIPE_DIR)] + [os.path.join(HERE, "..", "include", f) for f in ("rdvio_hip.h", "rdvio_pipeline.h")] + \
           [os.path.join(CSRC, "hypo_solvers.hpp"), LIB]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [os.environ.get("CXX", "g++")] + list(flags) + ["-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", "-o", out] + \
          [os.path.join(PIPE_DIR, s) for s in PIPE_SOURCES] + ["-L" + HERE, "-lrdvio_hip", "-Wl,-rpath," + HERE]
    subprocess.check_call(cmd)
    return out


EUROC_EXE = os.path.join(HERE, "test_euroc")


def build_test_euroc(force=False, verbose=False):
    """rd_vio_amd/test_euroc: the C++ EuRoC replay (host/test_euroc.cpp; the reference's examples/test_euroc.cpp headless)"""
    hdir = os.path.join(HERE, "host")
    deps = [os.path.join(hdir, f) for f in ("test_euroc.cpp", "rdvio_odometry.hpp", "rdvio_yaml.hpp", "rdvio_png.hpp")] + [PIPE_LIB, LIB]
    if not force and os.path.exists(EUROC_EXE) and all(os.path.getmtime(d) <= os.path.getmtime(EUROC_EXE) for d in deps):
        return EUROC_EXE
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-Wextra", "-ffp-contract=off", "-o", EUROC_EXE, os.path.join(hdir, "test_euroc.cpp"),
           "-L" + HERE, "-lrdvio_pipeline", "-lrdvio_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return EUROC_EXE


def build(force=False, verbose=False):
    lib = _build_hip(force, verbose)
    build_pipeline(force, verbose)
    build_test_euroc(force, verbose)
    return lib


def _build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DRDVIO_PROF"] if os.environ.get("RDVIO_PROF") else []  # diagnostic phase stamps (never timed builds)
    if os.environ.get("RDVIO_CHECK_UG"):
        extra.append("-DRDVIO_CHECK_UG")    # RDVIO_UG refuses LDS addresses and reports them at the next fetch (never a timed build)
    if os.environ.get("RDVIO_SOLVER_THREADS"):
        extra.append("-DRDVIO_SOLVER_THREADS=" + os.environ["RDVIO_SOLVER_THREADS"])   # experiment switch (block size of the solver kernels)
    if os.environ.get("RDVIO_PROF_CHOL"):
        extra.append("-DRDVIO_PROF_CHOL")   # stamps inside cholesky_lds (summary[72..76]; not together with RDVIO_PROF_HBLK)
    if os.environ.get("RDVIO_PROF_HBLK"):
        extra.append("-DRDVIO_PROF_HBLK")   # finer stamps inside the H-block pass (summary[72..75])
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        cmd = [hipcc] + FLAGS + extra + [f"-ffp-contract={CONTRACT.get(s, 'off')}", "-x", "hip", "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
