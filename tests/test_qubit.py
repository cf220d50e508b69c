"""Qubit / Pauli projection path.

CPU: the numpy oracle against golden vectors produced by the reference itself
(tests/golden/make_golden_qubit.py) and against the reference's literal answers
(test/test_qubit.py:108-147); the HIP kernel sources under the thread emulator.
GPU (-m gpu): the HIP path through the C ABI, bit-exact on rows/cols and exact on the
(+-1, +-i)-valued amplitudes."""
import ctypes
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import qubit_oracle as QO
from qiskit_addon_sqd_amd import _capi, qubit as Q

GOLD = json.loads((Path(__file__).parent / "golden" / "qubit_layer.json").read_text())
TERM_CASES = [k for k in GOLD if "label" in GOLD[k]]


def _term(label):
    t = Q.PauliTerm.from_label(label)
    return t.x, t.z


@pytest.mark.parametrize("key", TERM_CASES)
def test_oracle_matrix_elements_golden(key):
    g = GOLD[key]
    mat = np.array(g["matrix"], dtype=bool)
    amp, r, c = QO.matrix_elements_from_pauli(mat, *_term(g["label"]))
    assert r.tolist() == g["rows"] and c.tolist() == g["cols"]
    assert np.array_equal(np.real(amp), g["amp_re"]) and np.array_equal(np.imag(amp), g["amp_im"])


def test_oracle_reference_literals():
    # test/test_qubit.py:108-147
    mat = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], dtype=bool)
    amp, r, c = QO.matrix_elements_from_pauli(mat, *_term("XZ"))
    assert amp.tolist() == [1, -1, 1, -1] and r.tolist() == [0, 1, 2, 3] and c.tolist() == [2, 3, 0, 1]
    rows = ["0000", "0001", "0010", "0100", "0101", "1000", "1010"]
    mat = np.array([[ch == "1" for ch in s] for s in rows])
    amp, r, c = QO.matrix_elements_from_pauli(mat, *_term("XZIY"))
    assert amp.tolist() == [-1j, 1j] and r.tolist() == [1, 5] and c.tolist() == [5, 1]
    g = GOLD["sort_dedupe"]
    assert QO.sort_and_remove_duplicates(np.array(g["matrix"], dtype=bool)).astype(int).tolist() == g["out"]


def test_oracle_operator_golden():
    g = GOLD["operator_2local"]
    mat = np.array(g["matrix"], dtype=bool)
    terms = [(*_term(l), c) for l, c in zip(g["labels"], g["coeffs"])]
    dense = np.asarray(QO.project_operator_to_subspace(mat, terms).todense())
    assert np.allclose(dense, np.array(g["dense_re"]) + 1j * np.array(g["dense_im"]), atol=1e-13)


def _check_native_against_golden():
    for key in TERM_CASES:
        g = GOLD[key]
        mat = np.array(g["matrix"], dtype=bool)
        amp, r, c = Q.matrix_elements_from_pauli(mat, Q.PauliTerm.from_label(g["label"]))
        assert r.tolist() == g["rows"] and c.tolist() == g["cols"], key
        assert np.array_equal(np.real(amp), g["amp_re"]) and np.array_equal(np.imag(amp), g["amp_im"]), key
    g = GOLD["operator_2local"]
    mat = np.array(g["matrix"], dtype=bool)
    ham = Q.PauliSum.from_list(list(zip(g["labels"], g["coeffs"])))
    op = Q.project_operator_to_subspace(mat, ham)
    assert type(op).__name__ == g["type"] == "csr_matrix" and op.dtype == np.complex128
    ref = np.array(g["dense_re"]) + 1j * np.array(g["dense_im"])
    assert np.allclose(np.asarray(op.todense()), ref, atol=1e-13)
    e, v = Q.solve_qubit(mat, ham, k=1, which="SA")
    # the projected matrix follows the reference's (row = input, column = connected) convention, i.e. the
    # transpose of <r|H|c>; Hermitian, so the spectrum is that of the dense matrix
    assert abs(e[0] - np.linalg.eigvalsh(ref)[0]) < 1e-9
    g = GOLD["sort_dedupe"]
    assert Q.sort_and_remove_duplicates(np.array(g["matrix"], dtype=bool)).astype(int).tolist() == g["out"]
    with pytest.raises(ValueError, match="length < 64"):
        Q.matrix_elements_from_pauli(np.zeros((2, 64), dtype=bool), Q.PauliTerm.from_label("I" * 64))
    with pytest.raises(ValueError, match="strictly ascending"):
        Q.matrix_elements_from_pauli(np.array([[1, 0], [0, 1]], dtype=bool), Q.PauliTerm.from_label("XZ"))


def test_native_through_emulator(emu_lib, monkeypatch):
    monkeypatch.setattr(_capi, "_LIB", emu_lib)
    _check_native_against_golden()


@pytest.mark.gpu
def test_native_on_gpu(hip_lib):
    _check_native_against_golden()


@pytest.mark.gpu
def test_gpu_large_random_2local(hip_lib):
    """BASELINE config 5 shape at reduced size for the oracle: 40 qubits, random 2-local Hamiltonian,
    20 000 sorted unique bitstrings (+ partners), against the numpy restatement term by term."""
    rng = np.random.default_rng(5)
    nq, d = 40, 20000
    mat = rng.integers(2, size=(d, nq)).astype(bool)
    flips = np.zeros((200, nq), dtype=bool)
    for f in flips:
        f[rng.choice(nq, 2, replace=False)] = True
    extra = mat[rng.integers(d, size=4000)] ^ flips[rng.integers(200, size=4000)]
    mat = Q.sort_and_remove_duplicates(np.concatenate([mat, extra]))
    labels, coeffs = [], []
    for _ in range(60):
        i, j = rng.choice(nq, 2, replace=False)
        lab = ["I"] * nq
        lab[i], lab[j] = rng.choice(list("XYZ")), rng.choice(list("XYZ"))
        labels.append("".join(lab)); coeffs.append(float(rng.standard_normal()))
    ham = Q.PauliSum.from_list(list(zip(labels, coeffs)))
    op = Q.project_operator_to_subspace(mat, ham)
    ref = QO.project_operator_to_subspace(mat, [(*_term(l), c) for l, c in zip(labels, coeffs)]).tocsr()
    ref.sum_duplicates(); ref.eliminate_zeros(); ref.sort_indices()
    assert op.shape == ref.shape and op.nnz == ref.nnz
    assert np.array_equal(op.indptr, ref.indptr) and np.array_equal(op.indices, ref.indices)
    assert np.allclose(op.data, ref.data, atol=1e-12)
    assert abs(op - op.getH()).max() < 1e-12  # Hermitian


@pytest.mark.gpu
def test_gpu_config5_full_size(hip_lib):
    """BASELINE config 5 at its full size: 40 qubits, 1e5 sampled bitstrings (sorted, unique), a random 2-local
    Hamiltonian of 60 terms, the projected operator against the numpy restatement -- CSR bit for bit in its structure."""
    rng = np.random.default_rng(15)
    nq, d = 40, 100_000
    mat = rng.integers(2, size=(d, nq)).astype(bool)
    flips = np.zeros((400, nq), dtype=bool)
    for f in flips:
        f[rng.choice(nq, 2, replace=False)] = True
    extra = mat[rng.integers(d, size=20000)] ^ flips[rng.integers(400, size=20000)]  # partners, so that terms connect
    mat = Q.sort_and_remove_duplicates(np.concatenate([mat, extra]))[:d]
    assert mat.shape[0] == d
    labels, coeffs = [], []
    for _ in range(60):
        i, j = rng.choice(nq, 2, replace=False)
        lab = ["I"] * nq
        lab[i], lab[j] = rng.choice(list("XYZ")), rng.choice(list("XYZ"))
        labels.append("".join(lab)); coeffs.append(float(rng.standard_normal()))
    ham = Q.PauliSum.from_list(list(zip(labels, coeffs)))
    op = Q.project_operator_to_subspace(mat, ham)
    ref = QO.project_operator_to_subspace(mat, [(*_term(l), c) for l, c in zip(labels, coeffs)]).tocsr()
    ref.sum_duplicates(); ref.eliminate_zeros(); ref.sort_indices()
    assert op.shape == ref.shape == (d, d) and op.nnz == ref.nnz
    assert np.array_equal(op.indptr, ref.indptr) and np.array_equal(op.indices, ref.indices)
    assert np.allclose(op.data, ref.data, atol=1e-12)


def _check_large_scan_path():
    # d > 16384 rows exercises the three-phase (tile sums / scan / tile scan) CSR pointer construction
    rng = np.random.default_rng(11)
    nq = 20
    vals = np.unique(rng.integers(0, 1 << nq, size=21000, dtype=np.uint64))[:18000]
    mat = ((vals[:, None] >> np.arange(nq - 1, -1, -1, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    label = "XIZIIIIIIYIIIIIIIIZI"
    amp, r, c = Q.matrix_elements_from_pauli(mat, Q.PauliTerm.from_label(label))
    ramp, rr, rc = QO.matrix_elements_from_pauli(mat, *_term(label))
    assert np.array_equal(r, rr) and np.array_equal(c, rc) and np.array_equal(amp, ramp.astype(np.complex128))


def test_large_scan_path_emulator(emu_lib, monkeypatch):
    monkeypatch.setattr(_capi, "_LIB", emu_lib)
    _check_large_scan_path()


@pytest.mark.gpu
def test_large_scan_path_gpu(hip_lib):
    _check_large_scan_path()
