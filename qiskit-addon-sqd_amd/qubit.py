"""Drop-in for the qubit / Pauli path of the reference (``qiskit_addon_sqd/qubit.py``):
``solve_qubit`` (:29-75), ``project_operator_to_subspace`` (:78-144), ``sort_and_remove_duplicates``
(:147-164), ``matrix_elements_from_pauli`` (:167-240).  The projection runs in ``libsqd_hip.so``
(``csrc/sqd_pauli.hip``): XOR-connect, sign/phase, sorted lookup and the summation over terms in one
device pass pair, returning CSR.  No qiskit / jax import: an operator is anything that exposes
``.paulis`` (objects with little-endian bool arrays ``.x`` / ``.z``) and ``.coeffs`` -- e.g. a qiskit
``SparsePauliOp`` -- or this module's ``PauliSum`` built from labels.  The final ``eigsh`` stays scipy
on the host, as in the reference.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Sequence

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import eigsh

from . import _capi


@dataclass(frozen=True)
class PauliTerm:
    """One Pauli string; ``x`` / ``z`` are little-endian bool arrays (index q = qubit q), the layout of
    qiskit's ``Pauli.x`` / ``Pauli.z``."""

    x: np.ndarray
    z: np.ndarray

    @classmethod
    def from_label(cls, label: str) -> "PauliTerm":
        """``label[0]`` acts on the highest qubit (qiskit convention)."""
        chars = label[::-1]
        return cls(np.array([c in "XY" for c in chars]), np.array([c in "ZY" for c in chars]))


@dataclass(frozen=True)
class PauliSum:
    """Minimal stand-in for ``qiskit.quantum_info.SparsePauliOp``: ``paulis``, ``coeffs``, ``size``."""

    paulis: tuple
    coeffs: np.ndarray

    @classmethod
    def from_list(cls, terms: Sequence[tuple[str, complex]]) -> "PauliSum":
        return cls(tuple(PauliTerm.from_label(l) for l, _ in terms), np.array([c for _, c in terms], dtype=complex))

    @property
    def size(self) -> int:
        return len(self.paulis)


def _check_width(bitstring_matrix: np.ndarray) -> None:
    if bitstring_matrix.shape[1] > 63:
        raise ValueError("Bitstrings (rows) in bitstring_matrix must have length < 64.")


def _rows_to_uint64(bitstring_matrix: np.ndarray) -> np.ndarray:
    # column 0 is the most significant bit.  packbits gives the value left-aligned in ceil(n/8) bytes;
    # zero-extend to 8 big-endian bytes and shift the padding out (no d x n integer temporaries)
    d, n = bitstring_matrix.shape
    packed = np.packbits(np.asarray(bitstring_matrix, dtype=bool), axis=1, bitorder="big")
    nbytes = packed.shape[1]
    wide = np.zeros((d, 8), dtype=np.uint8)
    wide[:, 8 - nbytes :] = packed
    return wide.view(">u8").reshape(d).astype(np.uint64) >> np.uint64(8 * nbytes - n)


def _mask(bits: np.ndarray) -> int:
    return int(sum(1 << q for q in np.nonzero(np.asarray(bits))[0]))


def sort_and_remove_duplicates(bitstring_matrix: np.ndarray) -> np.ndarray:
    """Sort a bitstring matrix by unsigned integer value and drop repeated rows (``qubit.py:147-164``)."""
    _, indices = np.unique(_rows_to_uint64(bitstring_matrix), return_index=True)
    return bitstring_matrix[indices, :]


def _project(rows: np.ndarray, groups, device: int = 0):
    """groups: list of (xmask, [(zmask, coefficient incl. Y phase), ...]) -> CSR pieces + kernel ms."""
    lib = _capi.load_library()
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    d = rows.size
    xm = np.array([g[0] for g in groups], dtype=np.uint64)
    gptr = np.zeros(len(groups) + 1, dtype=np.int64)
    zs, cs = [], []
    for i, (_, terms) in enumerate(groups):
        gptr[i + 1] = gptr[i] + len(terms)
        for z, c in terms:
            zs.append(z)
            cs.append(c)
    zm = np.array(zs, dtype=np.uint64)
    coef = np.ascontiguousarray(np.array(cs, dtype=np.complex128)).view(np.float64)
    indptr = np.empty(d + 1, dtype=np.int64)
    nnz = C.c_int64()
    plan = C.c_void_p()
    p = lambda a, t: a.ctypes.data_as(t)
    rc = lib.sqd_pauli_count(device, p(rows, _capi._u64p), d, len(groups), p(xm, _capi._u64p), p(gptr, _capi._i64p),
                             p(zm, _capi._u64p), p(coef, _capi._dp), p(indptr, _capi._i64p), C.byref(nnz), C.byref(plan))
    if rc != 0:
        msg = lib.sqd_last_error().decode()
        raise (ValueError if rc == -1 else _capi.SQDNativeError)(msg)
    try:
        indices = np.empty(nnz.value, dtype=np.int64)
        data = np.empty(nnz.value, dtype=np.complex128)
        ms = C.c_double()
        rc = lib.sqd_pauli_fill(plan, p(indices, _capi._i64p), p(data.view(np.float64), _capi._dp), C.byref(ms))
        if rc != 0:
            raise _capi.SQDNativeError(lib.sqd_last_error().decode())
    finally:
        lib.sqd_pauli_free(plan)
    return indptr, indices, data, ms.value


def _term_masks(pauli):
    x, z = np.asarray(pauli.x, dtype=bool), np.asarray(pauli.z, dtype=bool)
    return _mask(x), _mask(z), (1j) ** int(np.count_nonzero(x & z))


def matrix_elements_from_pauli(bitstring_matrix: np.ndarray, pauli, *, device: int = 0):
    """Sparse matrix elements of one Pauli operator in the subspace (``qubit.py:167-240``): returns
    ``(amplitudes, row_indices, col_indices)`` with ``A[row, col] = amplitude``, rows ascending.  The
    bitstrings must be unique and sorted ascending (see ``sort_and_remove_duplicates``)."""
    _check_width(bitstring_matrix)
    xm, zm, phase = _term_masks(pauli)
    indptr, indices, data, _ = _project(_rows_to_uint64(bitstring_matrix), [(xm, [(zm, phase)])], device)
    rows = np.repeat(np.arange(bitstring_matrix.shape[0]), np.diff(indptr))
    return data, rows, indices


def project_operator_to_subspace(bitstring_matrix: np.ndarray, hamiltonian, *, verbose: bool = False, device: int = 0):
    """Project a Pauli-sum operator onto the subspace spanned by the (sorted, unique) bitstrings
    (``qubit.py:78-144``).  Returns a ``scipy.sparse.csr_matrix`` (complex128) -- what the reference's
    ``operator += coefficient * coo_matrix(...)`` accumulation yields in practice -- with rows = input
    configurations and columns = connected configurations."""
    _check_width(bitstring_matrix)
    d = bitstring_matrix.shape[0]
    by_x: dict[int, list] = {}
    for pauli, coeff in zip(hamiltonian.paulis, hamiltonian.coeffs):
        xm, zm, phase = _term_masks(pauli)
        by_x.setdefault(xm, []).append((zm, complex(coeff) * phase))
    if not by_x:
        return csr_matrix((d, d), dtype="complex128")
    groups = sorted(by_x.items())
    if verbose:  # pragma: no cover
        print(f"Projecting {len(hamiltonian.coeffs)} terms in {len(groups)} x-mask groups onto {d} states ...")
    indptr, indices, data, _ = _project(_rows_to_uint64(bitstring_matrix), groups, device)
    out = csr_matrix((data, indices, indptr), shape=(d, d), dtype="complex128")
    out.sort_indices()
    out.eliminate_zeros()
    return out


def solve_qubit(bitstring_matrix: np.ndarray, hamiltonian, *, verbose: bool = False, device: int = 0, **scipy_kwargs):
    """Energies and eigenstates of the Hamiltonian projected into the subspace (``qubit.py:29-75``)."""
    _check_width(bitstring_matrix)
    bitstring_matrix = sort_and_remove_duplicates(bitstring_matrix)
    ham_proj = project_operator_to_subspace(bitstring_matrix, hamiltonian, verbose=verbose, device=device)
    if verbose:  # pragma: no cover
        print("Diagonalizing Hamiltonian in the subspace...")
    return eigsh(ham_proj, **scipy_kwargs)
