"""CPU-side checks of the drop-in boundary: the library builds for gfx950, loads, and exports every
symbol include/rdvio_hip.h declares (no compute calls here -- there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

import rd_vio_amd
from rd_vio_amd import binding, build as rbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    rbuild.build()
    return rd_vio_amd.load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rdvio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdvio_hip_\w+)\s*\(", text)))


def test_header_symbols_all_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rdvio_hip.h but not exported"
    assert sorted(binding.EXPORTS) == declared


def test_version_and_layout(lib):
    assert b"gfx950" in lib.rdvio_hip_version()
    L = rd_vio_amd.PyrLayout()
    assert lib.rdvio_hip_pyr_layout_init(752, 480, 3, ctypes.byref(L)) == 0
    assert L.levels == 4 and list(L.w) == [752, 376, 188, 94] and list(L.h) == [480, 240, 120, 60]
    assert all(s % 64 == 0 for s in L.stride) and L.border == 32
    assert lib.rdvio_hip_pyr_layout_init(0, 480, 3, ctypes.byref(L)) != 0


def test_layout_matches_oracle(lib, oracle):
    for (w, h) in ((752, 480), (1280, 720), (100, 90), (64, 48)):
        L = rd_vio_amd.PyrLayout()
        lib.rdvio_hip_pyr_layout_init(w, h, 3, ctypes.byref(L))
        Lo = oracle.pyr_layout(w, h, 3)
        for f, _ in rd_vio_amd.PyrLayout._fields_:
            a, b = getattr(L, f), getattr(Lo, f)
            assert (list(a) == list(b)) if hasattr(a, "__len__") else (a == b), (w, h, f)


def test_code_object_is_gfx950():
    # the fat binary embedded in the .so must carry a gfx950 code object
    data = open(rd_vio_amd.lib_path(), "rb").read()
    assert b"gfx950" in data


@pytest.mark.skipif(rd_vio_amd.have_gpu(), reason="GPU present: context creation is covered by -m gpu tests")
def test_fails_loudly_without_gpu(lib):
    # no silent CPU fallback: without a device the product refuses to create a context
    with pytest.raises(rd_vio_amd.RdvioError):
        rd_vio_amd.Context()


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under rd_vio_amd/ may reference it
    pkg = os.path.join(ROOT, "rd_vio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f
                assert "#include \"rdvio_oracle.h\"" not in src and "ro_math.h" not in src, f


def test_sequence_driver_rejects_bad_arguments(lib):
    # the multi-sequence driver validates its arguments before it touches a device
    assert lib.rdvio_hip_frame_step(None, 0) != 0
    assert lib.rdvio_hip_run_sequences(None, 0, 0, 1, -1, None, None) != 0
    d = binding.FrameStep()
    assert lib.rdvio_hip_frame_step(ctypes.byref(d), 0) != 0          # no context
    assert lib.rdvio_hip_run_sequences(ctypes.byref(d), 1, 0, 0, -1, None, None) != 0   # zero steps
    assert lib.rdvio_hip_ctx_set_wait_mode(None, 1) != 0
