"""pytest configuration: ``gpu`` marker, import paths, native-library fixtures."""
import ctypes
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

EMU_DIR = ROOT / "tests" / "emu"
EMU_LIB = EMU_DIR / "_build" / "libsqd_emu.so"
CSRC = ROOT / "qiskit-addon-sqd_amd" / "csrc"
HIP_SOURCES = ["sqd_tables.hip", "sqd_sigma.hip", "sqd_lists.hip", "sqd_spmm.hip", "sqd_opp.hip", "sqd_oppsrc.hip", "sqd_davidson.hip", "sqd_rdm.hip", "sqd_pauli.hip", "sqd_recover.hip", "sqd_capi.hip"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _build_emu() -> Path:
    """g++ build of the UNMODIFIED kernel sources against tests/emu/hip/hip_runtime.h (logic checks only)."""
    srcs = [CSRC / s for s in HIP_SOURCES]
    deps = srcs + sorted(CSRC.glob("*.h")) + [ROOT / "include" / "sqd_hip.h", EMU_DIR / "hip" / "hip_runtime.h"]
    if EMU_LIB.exists() and all(d.stat().st_mtime <= EMU_LIB.stat().st_mtime for d in deps):
        return EMU_LIB
    EMU_LIB.parent.mkdir(exist_ok=True)
    cmd = ["g++", "-std=c++20", "-O1", "-fPIC", "-shared", "-pthread", "-Wno-psabi", f"-I{EMU_DIR}", f"-I{ROOT / 'include'}",
           f"-I{CSRC}", "-x", "c++", *map(str, srcs), "-o", str(EMU_LIB)]
    subprocess.run(cmd, check=True)
    return EMU_LIB


@pytest.fixture(scope="session")
def emu_lib():
    from qiskit_addon_sqd_amd import _capi

    return _capi.bind(ctypes.CDLL(str(_build_emu())))


@pytest.fixture(scope="session")
def hip_lib():
    from qiskit_addon_sqd_amd import _capi

    return _capi.load_library()


@pytest.fixture()
def emu_backend(emu_lib, monkeypatch):
    """Route the Python host layer through the emulator build (CPU logic tests of fermion.py)."""
    from qiskit_addon_sqd_amd import _capi, fermion

    fermion.clear_context_cache()
    monkeypatch.setattr(_capi, "_LIB", emu_lib)
    yield emu_lib
    fermion.clear_context_cache()
