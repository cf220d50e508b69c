"""numpy oracle for the qubit / Pauli projection path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates reference ``qiskit_addon_sqd/qubit.py`` step by step with numpy in place of jax:
``_int_conversion_from_bts_array`` (:280-297), ``_connected_elements_and_amplitudes_bool`` (:243-268),
``matrix_elements_from_pauli`` (:167-240), ``project_operator_to_subspace`` (:78-144),
``sort_and_remove_duplicates`` (:147-164).  Integer/bool work: bit-exact; pinned by the reference's
literal answers (test/test_qubit.py:70-157) and by tests/golden/qubit_layer.json generated from the
reference itself under stubs."""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix


def int_conversion(bitstring_matrix: np.ndarray) -> np.ndarray:
    n = bitstring_matrix.shape[1]
    out = np.zeros(bitstring_matrix.shape[0], dtype=np.int64)
    for i in range(n):
        out = out + bitstring_matrix[:, i].astype(np.int64) * 2 ** (n - 1 - i)
    return out.astype("longlong")


def sort_and_remove_duplicates(bitstring_matrix: np.ndarray) -> np.ndarray:
    _, indices = np.unique(int_conversion(bitstring_matrix), return_index=True)
    return bitstring_matrix[indices, :]


def matrix_elements_from_pauli(bitstring_matrix: np.ndarray, x: np.ndarray, z: np.ndarray):
    """x, z: little-endian bool arrays of the Pauli (qiskit ``Pauli.x`` / ``Pauli.z``)."""
    d = bitstring_matrix.shape[0]
    row_ids = np.arange(d)
    diag = np.logical_not(x)[::-1]
    sign = z[::-1]
    imag = np.logical_and(x, z)[::-1]
    int_rows = int_conversion(bitstring_matrix)
    conn = bitstring_matrix == diag
    amplitudes = np.prod((-1) ** np.logical_and(bitstring_matrix, sign) * np.array(1j, dtype="complex64") ** imag, axis=1)
    int_conn = int_conversion(conn)
    mask = np.isin(int_conn, int_rows, assume_unique=True, kind="sort")
    return amplitudes[mask], row_ids[mask], np.searchsorted(int_rows, int_conn[mask])


def project_operator_to_subspace(bitstring_matrix: np.ndarray, terms):
    """terms: iterable of (x, z, coefficient)."""
    d = bitstring_matrix.shape[0]
    op = coo_matrix((d, d), dtype="complex128")
    for x, z, coeff in terms:
        amp, r, c = matrix_elements_from_pauli(bitstring_matrix, x, z)
        op += coeff * coo_matrix((amp, (r, c)), (d, d))
    return op
