"""CPU tests of the host layer: golden vectors of the reference for the integer functions the package
re-implements, the C-ABI library's symbol table, and the Python API (fermion.py) driven through the
kernel-logic emulator (tests/emu) -- no GPU, no compute calls into libsqd_hip.so."""
import ctypes
import json
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi
from qiskit_addon_sqd_amd.counts import bitstring_matrix_to_integers
from qiskit_addon_sqd_amd.fermion import (SCIResult, SCIState, _check_ci_strs, bitstring_matrix_to_ci_strs,
                                          solve_fermion, solve_sci, solve_sci_batch)

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests" / "golden" / "integer_layer.json").read_text())


@pytest.mark.parametrize("key", [k for k in GOLD if k.startswith("b2i_")])
def test_bitstring_matrix_to_integers_golden(key):
    case = GOLD[key]
    out = bitstring_matrix_to_integers(np.array(case["matrix"], dtype=bool))
    assert [str(int(x)) for x in out] == case["out"]
    assert str(out.dtype) == case["dtype"]


@pytest.mark.parametrize("key", [k for k in GOLD if k.startswith("ci_")])
def test_bitstring_matrix_to_ci_strs_golden(key):
    case = GOLD[key]
    a, b = bitstring_matrix_to_ci_strs(np.array(case["matrix"], dtype=bool), open_shell=case["open_shell"])
    assert [str(int(x)) for x in a] == case["a"] and [str(int(x)) for x in b] == case["b"]
    if "dtype" in case:
        assert str(np.asarray(a).dtype) == case["dtype"]


def test_check_ci_strs_golden():
    ok = GOLD["check_ok"]
    oa, ob = _check_ci_strs((np.array(ok["a"]), np.array(ok["b"])))
    assert oa.tolist() == ok["out_a"] and ob.tolist() == ok["out_b"]
    for key in ("check_bad_up", "check_bad_dn"):
        case = GOLD[key]
        with pytest.raises(ValueError) as exc:
            _check_ci_strs((np.array(case["a"]), np.array(case["b"])))
        assert str(exc.value) == case["error"]


def test_library_loads_and_exports_every_declared_symbol():
    """libsqd_hip.so (hipcc, gfx950) must load on a GPU-less box and export every function that
    include/sqd_hip.h declares; no compute call is made here."""
    header = (ROOT / "include" / "sqd_hip.h").read_text()
    declared = set(re.findall(r"\b(sqd_[a-z0-9_]+)\s*\(", header)) - {"sqd_ctx"}
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    lib = _capi.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.sqd_abi_version() == 3


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_LIB", None)
    monkeypatch.setattr(_capi, "LIB_PATH", tmp_path / "libsqd_hip.so")
    with pytest.raises(_capi.SQDNativeError, match="no CPU fallback"):
        _capi.load_library()


def test_scistate_validation_and_roundtrip(tmp_path):
    with pytest.raises(ValueError, match="'amplitudes' shape must be"):
        SCIState(np.zeros((2, 3)), np.array([1, 2]), np.array([1, 2]), 2, (1, 1))
    st = SCIState(np.arange(6.0).reshape(2, 3), np.array([1, 2]), np.array([1, 2, 4]), norb=3, nelec=(1, 1))
    st.save(tmp_path / "s.npz")
    back = SCIState.load(tmp_path / "s.npz")
    assert np.array_equal(back.amplitudes, st.amplitudes) and np.array_equal(back.ci_strs_b, st.ci_strs_b)
    assert int(back.norb) == 3 and tuple(back.nelec) == (1, 1)
    with np.load(tmp_path / "s.npz") as data:  # reference npz schema (fermion.py:90-99)
        assert set(data.files) == {"amplitudes", "ci_strs_a", "ci_strs_b", "norb", "nelec"}
    with pytest.raises(NotImplementedError):
        st.rdm(rank=3)


def test_python_api_through_emulator(emu_backend):
    norb, nelec = 6, (3, 2)
    h1, eri = O.synthetic_integrals(norb, seed=3)
    sa = O.hf_centred_strings(norb, 3, 10, 1)
    sb = O.hf_centred_strings(norb, 2, 8, 2)
    e, state, occ, s2 = solve_fermion((sa, sb), h1, eri)
    e_ref, amps_ref, occ_ref, s2_ref, _ = O.solve_fermion_dense((sa, sb), h1, eri)
    assert abs(e - e_ref) < 1e-8 and abs(s2 - s2_ref) < 1e-6
    assert np.allclose(occ[0], occ_ref[0], atol=1e-6) and np.allclose(occ[1], occ_ref[1], atol=1e-6)
    assert state.nelec == (3, 2) and state.amplitudes.shape == (10, 8)
    # bitstring-matrix input path (left half = beta, right half = alpha), open shell
    from qiskit_addon_sqd_amd.synthetic import bitstring_matrix_from_strings

    mat = bitstring_matrix_from_strings(sa[:8], sb, norb)
    e2, st2, _, _ = solve_fermion(mat, h1, eri, open_shell=True)
    assert np.array_equal(st2.ci_strs_a, np.unique(sa[:8])) and np.array_equal(st2.ci_strs_b, sb)
    # solve_sci / solve_sci_batch: SCIResult with rdm1/rdm2 and energy from the RDMs (fermion.py:725-742)
    res = solve_sci((sa, sb), h1, eri, norb, nelec)
    assert isinstance(res, SCIResult) and abs(res.energy - e_ref) < 1e-8
    assert abs(np.trace(res.rdm1) - 5) < 1e-9 and res.rdm2.shape == (6,) * 4
    # default: RDMs are built on first access; compute_rdms=True builds them in the call and contracts
    # the energy from them exactly as the reference does -- same numbers either way
    lazy = solve_sci((sa, sb), h1, eri, norb, nelec)
    assert lazy._is_lazy() and "rdm2" not in lazy.__dict__
    eager = solve_sci((sa, sb), h1, eri, norb, nelec, compute_rdms=True)
    assert eager.__dict__.get("rdm2") is not None
    # the dataclass is the reference's: five fields (fermion.py:142-159), whatever the laziness
    import dataclasses
    assert [f.name for f in dataclasses.fields(SCIResult)] == ["energy", "sci_state", "orbital_occupancies", "rdm1", "rdm2"]
    assert len(dataclasses.astuple(lazy)) == 5 and dataclasses.replace(eager, energy=0.0).energy == 0.0
    with pytest.raises(NotImplementedError, match="lowest root"):
        solve_sci((sa, sb), h1, eri, norb, nelec, nroots=2)
    with pytest.raises(NotImplementedError, match="symmetry"):
        solve_sci((sa, sb), h1, eri, norb, nelec, wfnsym="A1g")
    assert abs(solve_sci((sa, sb), h1, eri, norb, nelec, nroots=1, pspace_size=400).energy - res.energy) < 1e-10
    with pytest.raises(ValueError, match="non-negative"):
        solve_sci((np.array([-3, 1, 2]), sb), h1, eri, norb, nelec)
    assert abs(eager.energy - res.energy) < 1e-10
    assert np.allclose(eager.rdm1, res.rdm1, atol=1e-12) and np.allclose(eager.rdm2, res.rdm2, atol=1e-12)
    assert np.allclose(res.sci_state.orbital_occupancies()[0], res.orbital_occupancies[0], atol=1e-12)
    with pytest.raises(ValueError, match="does not match"):
        solve_sci((sa, sb), h1, eri, norb, (2, 2))
    with pytest.raises(ValueError, match="hamming weight"):
        solve_fermion((np.array([7, 3]), sb), h1, eri)
    with pytest.raises(TypeError):
        solve_fermion((sa, sb), h1, eri, bogus_kwarg=1)
    out = solve_sci_batch([(sa, sb), (sa[:5], sb[:4])], h1, eri, norb, nelec, compute_rdms=False)
    assert len(out) == 2 and out[0].rdm2 is None and out[0].energy <= out[1].energy + 1e-9


def test_fcidump_round_trip(tmp_path):
    """The benchmark's "synthetic FCIDUMP" really goes through the file format: 8-fold unique
    entries written with 17 significant digits, expanded again on reading -- bit-for-bit."""
    from qiskit_addon_sqd_amd import synthetic as S

    for norb, nelec in ((2, 2), (7, 6)):
        h1, eri = S.synthetic_integrals(norb)
        path = tmp_path / f"fcidump_{norb}"
        S.write_fcidump(path, h1, eri, nelec, ms2=0, ecore=0.7137)
        h1r, erir, ne, ms2, ecore = S.read_fcidump(path)
        assert (ne, ms2, ecore) == (nelec, 0, 0.7137)
        assert np.array_equal(h1r, h1) and np.array_equal(erir, eri)
    head = (tmp_path / "fcidump_7").read_text().splitlines()[0]
    assert "NORB=7" in head and "NELEC=6" in head


def test_recover_configurations_deprecated_flat_occupancies():
    """reference configuration_recovery.py:100-108: a flat occupancy array [b_{N-1}..b_0, a_{N-1}..a_0] is still
    accepted (with a DeprecationWarning) and means the same as the (occ_a, occ_b) tuple."""
    from qiskit_addon_sqd_amd.sampling import recover_configurations

    rng = np.random.default_rng(3)
    norb = 4
    mat = rng.random((30, 2 * norb)) < 0.5
    probs = np.full(30, 1 / 30)
    occ_a, occ_b = rng.random(norb), rng.random(norb)
    ref_m, ref_p = recover_configurations(mat, probs, (occ_a, occ_b), 2, 2, rand_seed=11)
    flat = np.concatenate((np.flip(occ_b), np.flip(occ_a)))
    with pytest.warns(DeprecationWarning):
        m, p = recover_configurations(mat, probs, flat, 2, 2, rand_seed=11)
    assert np.array_equal(m, ref_m) and np.array_equal(p, ref_p)


def test_writeable_integrals_speculation_is_safe(emu_backend):
    """Plain (writeable) numpy integrals -- the reference user's call: the second call with the same array objects takes
    the previous call's solver context at once and hashes the bytes on a worker thread while the solve runs; an
    IN-PLACE edit of either tensor between calls must still be seen (the speculative result is discarded and the solve
    repeated on a fresh context), and going back to the old values finds the old context again."""
    from qiskit_addon_sqd_amd import fermion as F

    norb, nelec = 14, (3, 3)  # 14^4 elements: above the size below which every call hashes in full anyway
    h1, eri = O.synthetic_integrals(norb, seed=4)
    h1, eri = np.array(h1), np.array(eri)
    assert eri.flags.writeable and eri.size > F._HASH_SMALL
    sa, sb = O.hf_centred_strings(norb, 3, 12, 1), O.hf_centred_strings(norb, 3, 10, 2)

    def truth(h, e):  # frozen copies: identity-memoised contexts, no speculation
        return solve_fermion((sa, sb), *F.freeze_integrals(h.copy(), e.copy()))[0]

    e1 = solve_fermion((sa, sb), h1, eri)[0]
    assert e1 == truth(h1, eri)
    assert len(F._SPEC) == 1
    assert solve_fermion((sa, sb), h1, eri)[0] == e1  # speculative hit, digests agree
    eri *= 1.25  # in place: same object, same address
    e2 = solve_fermion((sa, sb), h1, eri)[0]
    assert e2 == truth(h1, eri) and abs(e2 - e1) > 1e-6
    h1[0, 0] -= 0.5
    e3 = solve_sci((sa, sb), h1, eri, norb, nelec).energy
    assert abs(e3 - truth(h1, eri)) < 1e-12 and abs(e3 - e2) > 1e-6
    h1[0, 0] += 0.5
    eri /= 1.25
    assert abs(solve_fermion((sa, sb), h1, eri)[0] - e1) < 1e-10


def test_native_digest_of_integral_tensors():
    """``sqd_hash_start`` / ``sqd_hash_finish`` (host code of the library: no device work): every byte counts, whatever
    the length class (tail bytes, whole stripes, several 512 KB pieces), jobs of several host threads may overlap, and the
    digest does not depend on the number of hash threads."""
    import subprocess
    import sys
    import threading

    from qiskit_addon_sqd_amd import fermion as F

    rng = np.random.default_rng(5)
    for nbytes in (0, 1, 7, 8, 63, 64, 65, 1000, 512 * 1024 - 1, 512 * 1024, 512 * 1024 + 9, 3 * 512 * 1024 + 77):
        a = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        d = F._native_digests(a, None)[0]
        assert F._native_digests(a.copy(), None)[0] == d
        for pos in {0, nbytes // 2, nbytes - 1} if nbytes else ():
            b = a.copy()
            b[pos] ^= 0x10
            assert F._native_digests(b, None)[0] != d, (nbytes, pos)
        if nbytes:
            assert F._native_digests(a[:-1].copy(), None)[0] != d  # the length is part of the digest
    big = rng.standard_normal(900_000)
    small = rng.standard_normal(900)
    want = F._native_digests(big, small)
    assert want[0] == F._native_digests(big, None)[0] and want[1] == F._native_digests(small, None)[0]
    # two jobs in flight at once, finished in the other order
    j1, j2 = F._native_start(big, small), F._native_start(small, big)
    assert F._native_finish(j2) == (want[1], want[0]) and F._native_finish(j1) == want
    got = []
    ts = [threading.Thread(target=lambda: got.append(F._native_digests(big, small))) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert got == [want] * 4
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from qiskit_addon_sqd_amd import fermion as F; "
            "rng = np.random.default_rng(5); [rng.integers(0, 256, size=n, dtype=np.uint8) for n in "
            "(0, 1, 7, 8, 63, 64, 65, 1000, 524287, 524288, 524297, 1572941)]; "
            "print(*F._native_digests(rng.standard_normal(900_000), rng.standard_normal(900)))" % str(ROOT))
    import os

    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "SQD_HASH_THREADS": "0"}, capture_output=True,
                         text=True, check=True).stdout.split()
    assert (int(out[0]), int(out[1])) == want
