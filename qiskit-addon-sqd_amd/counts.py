"""Bitstring helpers on the hot path's input side.

Mirrors the part of reference ``qiskit_addon_sqd/counts.py`` that
``bitstring_matrix_to_ci_strs`` needs (``counts.py:186-201``).  Pure integer work on the host;
bit-exact with the reference (pinned by ``tests/golden`` fixtures).
"""

from __future__ import annotations

import numpy as np


def bitstring_matrix_to_integers(bitstring_matrix: np.ndarray) -> np.ndarray:
    """Convert a bitstring matrix to an array of integers (column 0 is the most significant bit).

    Same contract as the reference (``counts.py:186-201``): ``int64`` results below 64 bits, Python
    object integers from 64 bits up.  Implemented as a packed dot product instead of a per-column
    Python loop.
    """
    bitstring_matrix = np.asarray(bitstring_matrix)
    n_bitstrings, n_bits = bitstring_matrix.shape
    if n_bits < 64:
        weights = np.left_shift(np.int64(1), np.arange(n_bits - 1, -1, -1, dtype=np.int64))
        return bitstring_matrix.astype(np.int64) @ weights if n_bits else np.zeros(n_bitstrings, dtype=int)
    result = np.zeros(n_bitstrings, dtype=object)
    mat = bitstring_matrix.astype(object)
    for i in range(n_bits):
        result += mat[:, i] * (1 << (n_bits - 1 - i))
    return result
