"""Host-side sample processing on either side of the solver seam.

Behavioural restatement -- same results AND the same consumption of the numpy ``Generator`` stream --
of the reference's ``counts.bit_array_to_arrays`` (``counts.py:45-61``),
``subsampling.postselect_by_hamming_right_and_left`` / ``subsample`` (``subsampling.py:96-211``) and
``configuration_recovery.recover_configurations`` (``configuration_recovery.py:59-306``).  It is pinned
by ``tests/golden/sqd_loop.json`` (the reference's own loop, run under import stubs, with every list of
CI strings it hands to ``sci_solver`` recorded) and by the reference's literal known answers.

The random choices stay numpy's (``Generator.choice(p=, replace=False)``): bit-for-bit reproducibility
of a seeded SQD run against the reference requires the identical stream, so this is host work by
design, not a kernel.  What is restructured: flip weights depend only on (orbital, bit value), so they
are computed once per call instead of once per bitstring; rows that already have the right Hamming
weights are never visited; and the per-row ``choice`` calls are *replayed* natively
(``csrc/sqd_recover.hip: sqd_recover_rows``) on a block of uniforms drawn from the same generator, which
is then rewound by what was not consumed -- 1e5 samples per iteration: 7.7 s -> 0.075 s, same rows,
same probabilities, same stream position (``tests/test_sqd_loop.py``).  Row deduplication works on
bit-packed rows (``np.unique`` on byte strings) instead of ``np.unique(axis=0)`` / a Python dict.
"""

from __future__ import annotations

from typing import Sequence

import warnings

import numpy as np


# ----------------------------------------------------------------------------- counts
def _row_keys(bools: np.ndarray) -> np.ndarray:
    """One sortable key per row of a bool matrix whose order is the row-wise lexicographic order of the
    bools (False < True): rows are packed MSB-first into bytes; up to 64 bits the bytes are read as one
    big-endian integer (integer sorts are several times faster), beyond that compared as byte strings."""
    packed = np.packbits(bools, axis=1)
    nbytes = packed.shape[1]
    if nbytes <= 8:
        wide = np.zeros((packed.shape[0], 8), dtype=np.uint8)
        wide[:, :nbytes] = packed  # left-aligned: trailing zero bytes do not change the order
        return wide.view(">u8").ravel().astype(np.uint64)
    packed = np.ascontiguousarray(packed)
    return packed.view(np.dtype((np.void, nbytes))).ravel()


def _unique_rows(bools: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """``np.unique(bools, axis=0, return_counts=True)`` for a bool matrix, an order of magnitude faster
    (sorts one key per row, see ``_row_keys``, instead of comparing rows column by column)."""
    if bools.ndim != 2 or bools.shape[0] == 0 or bools.shape[1] == 0:
        return np.unique(bools, axis=0, return_counts=True)
    if bools.dtype == np.bool_ and bools.shape[1] <= 128 and bools.shape[0] >= 4096:
        lib, capi = _native_lib()  # packed keys + radix sort natively (``sqd_unique_rows``); numpy below otherwise
        if lib is not None:
            rows = np.ascontiguousarray(bools)
            n = rows.shape[0]
            first, counts, nu = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64), capi.C.c_int64(0)
            if lib.sqd_unique_rows(rows.ctypes.data, n, rows.shape[1], first.ctypes.data, counts.ctypes.data, capi.C.byref(nu)) == 0:
                return rows[first[: nu.value]], counts[: nu.value].copy()
    _, first, counts = np.unique(_row_keys(bools), return_index=True, return_counts=True)
    return bools[first], counts


def bool_matrix_to_arrays(bool_array: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Unique rows (sorted as ``np.unique(axis=0)`` sorts them) and their empirical probabilities."""
    bool_array = np.asarray(bool_array, dtype=bool)
    bitstrings, counts = _unique_rows(bool_array)
    return bitstrings, counts / bool_array.shape[0]


def bit_array_to_arrays(bit_array) -> tuple[np.ndarray, np.ndarray]:
    """qiskit ``BitArray`` (anything with ``.array``, ``.num_bits``, ``.num_shots``) or a plain bool
    matrix -> (bitstring matrix, probabilities); reference ``counts.py:45-61``."""
    if hasattr(bit_array, "num_bits") and hasattr(bit_array, "array"):
        bools = np.unpackbits(bit_array.array, axis=-1)[..., -bit_array.num_bits :].astype(bool)
        bitstrings, counts = _unique_rows(bools.reshape(-1, bools.shape[-1]))
        return bitstrings, counts / bit_array.num_shots
    return bool_matrix_to_arrays(bit_array)


def counts_to_arrays(counts) -> tuple[np.ndarray, np.ndarray]:
    """Counts dictionary -> (bitstring matrix, probabilities); reference ``counts.py:24-42``."""
    if not counts:
        return np.array([]), np.array([])
    total = sum(counts.values())
    mat = np.array([[bit == "1" for bit in bitstring] for bitstring in counts])
    return mat, np.array([c / total for c in counts.values()])


# ----------------------------------------------------------------------------- post-selection / subsampling
def postselect_by_hamming_right_and_left(
    bitstring_matrix: np.ndarray, probabilities: np.ndarray, *, hamming_right: int, hamming_left: int
) -> tuple[np.ndarray, np.ndarray]:
    """Keep the bitstrings whose right / left halves have the requested Hamming weights and renormalise
    their probabilities (reference ``subsampling.py:96-144``, same error messages)."""
    if hamming_left < 0 or hamming_right < 0:
        raise ValueError("Hamming weight must be specified with a non-negative integer.")
    n_bitstrings, n_bits = bitstring_matrix.shape
    if n_bits % 2:
        raise ValueError(f"The length of the bitstrings must be even. Instead, got {n_bits}.")
    if len(probabilities) != n_bitstrings:
        raise ValueError(
            "The number of elements in the probabilities array must match the number of rows in the bitstring matrix."
        )
    norb = n_bits // 2
    keep = (np.sum(bitstring_matrix[:, norb:], axis=1) == hamming_right) & (
        np.sum(bitstring_matrix[:, :norb], axis=1) == hamming_left
    )
    probs = probabilities[keep]
    probs /= np.sum(probs)
    return bitstring_matrix[keep], probs


def subsample(
    bitstring_matrix: np.ndarray,
    probabilities: np.ndarray,
    samples_per_batch: int,
    num_batches: int,
    rand_seed: np.random.Generator | int | None = None,
) -> list[np.ndarray]:
    """``num_batches`` batches, each drawn without replacement with the given probabilities
    (reference ``subsampling.py:147-211``; one ``Generator.choice`` per batch, same arguments)."""
    if bitstring_matrix.shape[0] < 1:
        return [np.array([])] * num_batches
    if len(probabilities) != bitstring_matrix.shape[0]:
        raise ValueError(
            "The number of elements in the probabilities array must match the number of rows in the bitstring matrix."
        )
    if samples_per_batch < 1:
        raise ValueError("Samples per batch must be specified with a positive integer.")
    if num_batches < 1:
        raise ValueError("The number of batches must be specified with a positive integer.")
    rng = np.random.default_rng(rand_seed)
    n = bitstring_matrix.shape[0]
    if samples_per_batch >= n:
        return [bitstring_matrix[np.arange(n)] for _ in range(num_batches)]
    pool = np.arange(n).astype("int")
    idx_all = _choice_native(rng, probabilities, samples_per_batch, num_batches)
    if idx_all is not None:
        return [bitstring_matrix[idx] for idx in idx_all]
    return [
        bitstring_matrix[rng.choice(pool, samples_per_batch, replace=False, p=probabilities)] for _ in range(num_batches)
    ]


def _native_lib():
    """The native library when it can be loaded (the host-side helpers need no GPU), else None: numpy then."""
    try:
        from . import _capi

        return _capi.load_library(), _capi
    except Exception:  # noqa: BLE001
        return None, None


def _rewind(bitgen, before: dict, count: int) -> None:
    """Step a PCG64 back by ``count`` doubles.  ``advance()`` also clears the cached 32-bit half numpy keeps for 32-bit
    draws; numpy's own ``choice`` / ``random`` draw doubles only and leave it alone, so it is put back as it was."""
    bitgen.advance(-int(count))
    after = bitgen.state
    after["has_uint32"], after["uinteger"] = before["has_uint32"], before["uinteger"]
    bitgen.state = after


_SQD_ERR_LIMIT = -4  # include/sqd_hip.h: the block of uniforms was too short


def _choice_native(rng, probabilities, size: int, nbatches: int):
    """``nbatches`` calls ``rng.choice(n, size, replace=False, p=probabilities)`` replayed natively (``sqd_choice_replay``) on
    uniforms drawn from the same generator, which is rewound by what was not consumed: the indices and the stream position
    are numpy's (7 ms per batch at 1e5 samples in numpy -- validation passes and temporaries around a cumsum and ``size``
    binary searches; the batches of an iteration share the first round's cumulative sum here).  Returns [nbatches][size]
    indices, or None: not applicable (another bit generator, inputs numpy raises on) -- numpy then decides."""
    bitgen = rng.bit_generator
    if not isinstance(bitgen, np.random.PCG64):
        return None
    lib, capi = _native_lib()
    if lib is None:
        return None
    p = np.ascontiguousarray(probabilities, dtype=np.float64)
    if p.ndim != 1 or size > p.size:
        return None
    # A batch takes `size` doubles plus a few per round with two draws on one element.  The block of uniforms is drawn
    # with a modest slack first (1.25 size + 64 per batch: 1 MB for ten batches of 1e4, not the 4 size + 64 worst case up
    # front) and once more at the worst case if the replay ran out; beyond that: numpy's turn.
    state = bitgen.state
    out = np.empty((int(nbatches), int(size)), dtype=np.int64)
    used = capi.C.c_int64(0)
    small = int(size) * (int(size) + 1) // 2
    for per_batch in ((small,) if size < 64 else (int(size) + int(size) // 4 + 64, 4 * int(size) + 64)):
        bound = int(nbatches) * per_batch
        uniforms = rng.random(bound)
        rc = lib.sqd_choice_replay(p.ctypes.data, p.size, int(size), int(nbatches), uniforms.ctypes.data, uniforms.size,
                                   out.ctypes.data, capi.C.byref(used))
        if rc == 0:
            _rewind(bitgen, state, bound - used.value)
            return out
        bitgen.state = state
        if rc != _SQD_ERR_LIMIT:  # inputs numpy raises on (SQD_ERR_STATE): a longer block cannot help -- numpy's turn
            return None
        # SQD_ERR_LIMIT: the block was too short (many collisions) -- once more at the worst case
    return None


# ----------------------------------------------------------------------------- configuration recovery
def _flip_weight_up(ratio: float, occ: np.ndarray, eps: float = 0.01) -> np.ndarray:
    """Weight for turning a 0 into a 1 given the expected filling ``ratio`` and the orbital occupancy
    (piecewise linear, reference ``configuration_recovery.py:131-159``), for all orbitals at once."""
    occ = np.asarray(occ, dtype=float)
    below = occ * eps / ratio if ratio != 0 else np.zeros_like(occ)
    if ratio == 1.0:
        above = np.full_like(occ, eps)
    else:
        slope = (1 - eps) / (1 - ratio)
        above = occ * slope + (1 - slope)
    return np.where(occ < ratio, below, above)


def _flip_weight_down(ratio: float, occ: np.ndarray, eps: float = 0.01) -> np.ndarray:
    """Weight for turning a 1 into a 0 (``configuration_recovery.py:162-178``)."""
    return _flip_weight_up(1 - ratio, 1 - np.asarray(occ, dtype=float), eps)


def _repair_half(bits: np.ndarray, w_up: np.ndarray, w_down: np.ndarray, target: int, rng) -> None:
    """Bring one half of a bitstring (a view, modified in place) to Hamming weight ``target`` by
    flipping bits drawn without replacement with the occupancy-informed weights
    (``configuration_recovery.py:230-304``: same ``Generator.choice`` call, same arguments)."""
    weights = np.where(bits, w_down, w_up)
    weights = np.minimum(1, np.maximum(0, weights))
    if not np.any(weights):
        return
    weights /= np.sum(weights)
    excess = np.sum(bits) - target
    if excess > 0:
        candidates = np.where(bits)[0]
        p = weights[bits] / np.sum(weights[bits])
        bits[rng.choice(candidates, size=round(excess), replace=False, p=p)] = False
    elif excess < 0:
        empty = np.logical_not(bits)
        candidates = np.where(empty)[0]
        p = weights[empty] / np.sum(weights[empty])
        bits[rng.choice(candidates, size=round(np.abs(excess)), replace=False, p=p)] = True


def recover_configurations(
    bitstring_matrix: np.ndarray,
    probabilities: Sequence[float] | np.ndarray,
    avg_occupancies: tuple[np.ndarray, np.ndarray],
    num_elec_a: int,
    num_elec_b: int,
    rand_seed: np.random.Generator | int | None = None,
) -> tuple[np.ndarray, np.ndarray]:
    """Refine bitstrings toward the target Hamming weights using the average orbital occupancies
    (reference ``configuration_recovery.py:59-128``).  Bit ``i`` (< norb) is the spin-down partner of
    bit ``i + norb``; ``avg_occupancies = (occ_a, occ_b)`` indexed by orbital (LSB = rightmost bit)."""
    rng = np.random.default_rng(rand_seed)
    bitstring_matrix = np.asarray(bitstring_matrix, dtype=bool)
    norb = bitstring_matrix.shape[1] // 2
    if len(np.array(avg_occupancies).shape) == 1:  # deprecated flat layout (configuration_recovery.py:100-108)
        warnings.warn(
            "Passing avg_occupancies as a 1D array is deprecated. Pass a length-2 tuple containing the spin-up and spin-down occupancies respectively.",
            DeprecationWarning,
            stacklevel=2,
        )
        avg_occupancies = (np.flip(avg_occupancies[norb:]), np.flip(avg_occupancies[:norb]))
    if num_elec_a < 0 or num_elec_b < 0:
        raise ValueError("The numbers of electrons must be specified as non-negative integers.")
    # column c of the left half is beta orbital norb-1-c; of the right half alpha orbital norb-1-c
    occs = np.flip(np.asarray(avg_occupancies)).flatten()
    occ_left, occ_right = occs[:norb], occs[norb:]
    up_l, dn_l = _flip_weight_up(num_elec_b / norb, occ_left), _flip_weight_down(num_elec_b / norb, occ_left)
    up_r, dn_r = _flip_weight_up(num_elec_a / norb, occ_right), _flip_weight_down(num_elec_a / norb, occ_right)

    out = np.ascontiguousarray(bitstring_matrix).copy()
    probabilities = np.asarray(probabilities, dtype=float)
    fast = _recover_all_native(out, probabilities, norb, up_l, dn_l, up_r, dn_r, num_elec_b, num_elec_a, rng)
    if fast is not None:
        return fast
    out = np.ascontiguousarray(bitstring_matrix).copy()  # (a failed native pass may have touched some rows)
    sum_l, sum_r = out[:, :norb].sum(axis=1), out[:, norb:].sum(axis=1)
    rows = np.nonzero((sum_l != num_elec_b) | (sum_r != num_elec_a))[0]
    if rows.size and not _recover_rows_native(out, rows, sum_l, sum_r, norb, up_l, dn_l, up_r, dn_r,
                                              num_elec_b, num_elec_a, rng):
        out[rows] = bitstring_matrix[rows]  # (a failed native pass may have touched some rows)
        for i in rows:  # left (beta) half first, then right (alpha): the reference's stream order
            _repair_half(out[i, :norb], up_l, dn_l, num_elec_b, rng)
            _repair_half(out[i, norb:], up_r, dn_r, num_elec_a, rng)

    # merge duplicates in first-occurrence order, adding their probabilities in row order
    _, first_idx, inverse = np.unique(_row_keys(out), return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")  # groups in order of first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    # np.bincount adds the weights one by one in row order: the same sums as a running dict
    freqs = np.bincount(rank[inverse.ravel()], weights=probabilities, minlength=order.size)
    freqs = np.abs(freqs) / np.sum(np.abs(freqs))
    return out[first_idx[order]], freqs


def _recover_all_native(out, probabilities, norb, up_l, dn_l, up_r, dn_r, target_l, target_r, rng):
    """The whole of ``recover_configurations`` behind the flip weights in three native passes over the byte matrix --
    Hamming excess of every row (``sqd_hamming_excess``: the bound on the uniforms to draw), repair of the rows off target
    in row order (``sqd_recover_rows`` with rows = NULL), duplicate merge in first-occurrence order with the
    probabilities added in row order (``sqd_merge_rows``) -- instead of numpy's column sums, ``np.unique`` with inverse,
    ``argsort`` and ``bincount`` over 1e5 rows.  Returns (bitstrings, frequencies), or None when the fast path does not
    apply (another bit generator, > 64 orbitals, weights numpy would raise on): the stream is then untouched."""
    bitgen = rng.bit_generator
    if not isinstance(bitgen, np.random.PCG64) or norb > 64 or 2 * norb > 128 or out.shape[0] == 0:
        return None
    if probabilities.ndim != 1 or probabilities.size != out.shape[0]:
        return None
    lib, capi = _native_lib()
    if lib is None:
        return None
    C = capi.C
    work = out.view(np.uint8)
    n = out.shape[0]
    bound, nbad = C.c_int64(0), C.c_int64(0)
    if lib.sqd_hamming_excess(work.ctypes.data, n, int(norb), int(target_l), int(target_r), C.byref(bound), C.byref(nbad)) != 0:
        return None
    if nbad.value:
        state = bitgen.state
        uniforms = rng.random(bound.value)
        used = C.c_int64(0)
        args = [np.ascontiguousarray(a, dtype=np.float64) for a in (up_l, dn_l, up_r, dn_r)]
        dp, u8p = capi._dp, C.POINTER(C.c_uint8)
        rc = lib.sqd_recover_rows(work.ctypes.data_as(u8p), n, int(norb), None, n, *(a.ctypes.data_as(dp) for a in args),
                                  int(target_l), int(target_r), uniforms.ctypes.data_as(dp), uniforms.size, C.byref(used))
        if rc != 0:
            bitgen.state = state
            return None
        _rewind(bitgen, state, bound.value - used.value)
    probs = np.ascontiguousarray(probabilities, dtype=np.float64)
    first = np.empty(n, dtype=np.int64)
    freq = np.empty(n, dtype=np.float64)
    nu = C.c_int64(0)
    if lib.sqd_merge_rows(work.ctypes.data, n, int(2 * norb), probs.ctypes.data, first.ctypes.data, freq.ctypes.data,
                          C.byref(nu), 1) != 0:
        raise RuntimeError("sqd_merge_rows failed")  # (cannot happen behind the checks above; the stream has moved)
    freq = freq[: nu.value]
    freq = np.abs(freq) / np.sum(np.abs(freq))
    return out[: nu.value], freq  # (the distinct rows were moved to the front of `out`, a private copy)


def _recover_rows_native(out, rows, sum_l, sum_r, norb, up_l, dn_l, up_r, dn_r, target_l, target_r, rng) -> bool:
    """Repair ``rows`` of ``out`` in place through ``sqd_recover_rows`` (csrc/sqd_recover.hip), which replays
    numpy's ``Generator.choice(p=, replace=False)`` on a block of uniforms drawn here; the generator is
    rewound by what was not used, so the stream ends exactly where the per-row loop would leave it.
    Returns False (stream untouched; the caller restores the rows) when the fast path does not apply: a bit
    generator that cannot be rewound, more than 64 orbitals, or weights numpy would raise on."""
    bitgen = rng.bit_generator
    if not isinstance(bitgen, np.random.PCG64) or norb > 64:
        return False
    from . import _capi

    lib = _capi.load_library()
    ex_l = np.abs(sum_l[rows].astype(np.int64) - target_l)
    ex_r = np.abs(sum_r[rows].astype(np.int64) - target_r)
    # a call for k bits draws k, then at most k-1, ... uniforms: k(k+1)/2 bounds it
    bound = int((ex_l * (ex_l + 1) // 2 + ex_r * (ex_r + 1) // 2).sum())
    state = bitgen.state
    uniforms = rng.random(bound)
    work = out.view(np.uint8)
    rows64 = np.ascontiguousarray(rows, dtype=np.int64)
    used = _capi.C.c_int64(0)
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (up_l, dn_l, up_r, dn_r)]
    dp, u8p, i64p = _capi._dp, _capi.C.POINTER(_capi.C.c_uint8), _capi._i64p
    rc = lib.sqd_recover_rows(work.ctypes.data_as(u8p), out.shape[0], int(norb), rows64.ctypes.data_as(i64p),
                              rows64.size, *(a.ctypes.data_as(dp) for a in args), int(target_l), int(target_r),
                              uniforms.ctypes.data_as(dp), uniforms.size, _capi.C.byref(used))
    if rc != 0:  # the caller restores the rows from its input; the stream goes back to where it was
        bitgen.state = state
        return False
    _rewind(bitgen, state, bound - used.value)
    return True
