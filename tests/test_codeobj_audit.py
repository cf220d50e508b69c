"""Audit of the gfx950 code objects inside libsqd_hip.so (CPU; needs only the ROCm LLVM tools of the build image).

Round 5 shipped ``k_sigma<16, ., true, false>`` as a stub that CALLS an out-of-line ``sigma_body`` -- 35 000 lines of
assembly, whose long branches are materialised in ``s[30:31]``, the callee's return address: every wavefront ended up
spinning on the function epilogue and the launch never returned (``profiles/r06/hang_root_cause.txt``).  The 64-thread
emulator compiles the sources with g++ and cannot see a code-generation defect, so the shipped binary itself is checked:

* no device code of the library contains a call (``s_swappc_b64``) -- every ``__device__`` function is inlined, no
  kernel has a return address that branch relaxation could clobber;
* no kernel uses a dynamic stack;
* the sigma and opposite-spin kernels own no private segment (no spills, no stack frame).
"""
import re
import subprocess
import tempfile
from pathlib import Path

import pytest

import _codeobj

ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "qiskit-addon-sqd_amd" / "csrc" / "libsqd_hip.so"

pytestmark = pytest.mark.skipif(not (_codeobj.tools_available() and LIB.exists()), reason="needs the ROCm LLVM tools and a built library")


@pytest.fixture(scope="module")
def records():
    return _codeobj.kernel_records(LIB)


def test_every_kernel_is_listed(records):
    names = {r.get("demangled", r["name"]).split("(")[0] for r in records}
    for expected in ("sqd::k_sigma_direct<false>", "sqd::k_dots_s", "sqd::k_observables", "sqd::k_opp_reduce"):
        assert any(n.endswith(expected) or expected in n for n in names), expected
    assert len(records) > 150


def test_no_device_code_calls_out_of_line():
    offenders = []
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(_codeobj.code_objects(LIB)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(co)
            dis = subprocess.run([str(_codeobj.LLVM_BIN / "llvm-objdump"), "-d", "--no-show-raw-insn", str(f)],
                                 check=True, capture_output=True, text=True).stdout  # fmt: skip
            func = "?"
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    func = m.group(1)
                elif "s_swappc_b64" in line or "s_call_b64" in line:
                    offenders.append(func)
    assert not offenders, f"out-of-line device calls in: {sorted(set(offenders))[:8]}"


def test_no_dynamic_stack(records):
    bad = [r.get("demangled", r["name"]) for r in records if r["dynamic_stack"]]
    assert not bad, bad[:8]


def test_sigma_kernels_own_no_private_segment(records):
    pat = re.compile(r"sqd::k_(sigma|sigma_b|opp_rows|opp_src|spmm_grouped|sigma_lists|alpha_rows|same_spin_mfma)\b")
    allowed = {"sqd::k_sigma_rows<2, true>", "sqd::k_sigma_rows<2, false>"}  # (7-8 spilled registers, no calls; rows kernel of uniform 1000-5000 sets)
    bad = []
    for r in records:
        d = r.get("demangled", r["name"])
        if pat.search(d) and r["scratch"] > 0 and not any(d.startswith("void " + a) or d.startswith(a) for a in allowed):
            bad.append((d[:80], r["scratch"], r["vgpr_spill"]))
    assert not bad, bad
