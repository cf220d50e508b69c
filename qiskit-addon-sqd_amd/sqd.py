"""The configuration-recovery loop around the solver seam, free of qiskit / pyscf / jax imports.

Drop-in for reference ``qiskit_addon_sqd.fermion.diagonalize_fermionic_hamiltonian``
(``fermion.py:204-462``) with the same signature, validation messages, random-stream consumption
and result, and with this package's HIP-backed ``solve_sci_batch`` as the default ``sci_solver``.
``bit_array`` may be a qiskit ``BitArray`` (duck-typed) or a plain bool matrix of samples.

Per iteration (reference ``:415-459``): build the batches of CI strings (post-selection or
configuration recovery, subsampling, ordering by frequency with requested and carry-over strings
first, truncation, sorting -- ``_prepare_ci_strings`` ``:472-560``), hand the whole list to
``sci_solver`` (the only collective step), then pick the lowest-energy batch, test convergence
against the previous iteration and extract the carry-over strings
(``_process_sci_results`` ``:563-640``).  Pinned by ``tests/golden/sqd_loop.json``.
"""

from __future__ import annotations

from typing import Callable

import numpy as np

from .counts import bitstring_matrix_to_integers
from .fermion import SCIResult, freeze_integrals, solve_sci_batch
from .sampling import bit_array_to_arrays, postselect_by_hamming_right_and_left, recover_configurations, subsample


def _first_occurrences(vals: np.ndarray) -> np.ndarray:
    """Unique values in order of first appearance."""
    _, idx = np.unique(vals, return_index=True)
    idx.sort()
    return vals[idx]


def _by_descending_count(strings: np.ndarray, counts: np.ndarray) -> np.ndarray:
    return strings[np.argsort(counts)[::-1]]


def _batch_strings(samples, norb, symmetrize_spin, include_a, include_b, carry_a, carry_b, max_dim_a, max_dim_b):
    """One batch of sampled bitstrings -> sorted (alpha, beta) CI strings: requested strings first, then
    carry-over, then sampled half-strings by descending frequency; first occurrences; truncated.

    The order only decides what a truncation drops: without ``max_dim`` the result is the sorted union of the three
    sources (one ``np.unique`` per spin instead of five sorts)."""
    ints_a = bitstring_matrix_to_integers(samples[:, norb:])
    ints_b = bitstring_matrix_to_integers(samples[:, :norb])
    if symmetrize_spin and max_dim_a is None:
        merged = np.unique(np.concatenate((include_a, include_b, carry_a, ints_a, ints_b)))
        return merged, merged
    if not symmetrize_spin and max_dim_a is None and max_dim_b is None:
        return np.unique(np.concatenate((include_a, carry_a, ints_a))), np.unique(np.concatenate((include_b, carry_b, ints_b)))
    sa, ca = np.unique(ints_a, return_counts=True)
    sb, cb = np.unique(ints_b, return_counts=True)
    if symmetrize_spin:
        pooled = _by_descending_count(np.concatenate((sa, sb)), np.concatenate((ca, cb)))
        merged = _first_occurrences(np.concatenate((include_a, include_b, carry_a, pooled)))[:max_dim_a]
        strs_a = strs_b = merged
    else:
        strs_a = _first_occurrences(np.concatenate((include_a, carry_a, _by_descending_count(sa, ca))))[:max_dim_a]
        strs_b = _first_occurrences(np.concatenate((include_b, carry_b, _by_descending_count(sb, cb))))[:max_dim_b]
    strs_a.sort()
    strs_b.sort()
    return strs_a, strs_b


def _carryover(result: SCIResult, threshold: float, symmetrize_spin: bool):
    """Strings whose determinants carry |amplitude| >= threshold, by descending marginal weight (reference
    ``fermion.py:607-631``; the rows / columns holding such an amplitude come from one mask instead of a sort of all D)."""
    state = result.sci_state
    amps = state.amplitudes
    big = np.abs(amps) >= threshold
    ia, ib = np.flatnonzero(big.any(axis=1)), np.flatnonzero(big.any(axis=0))
    keep_a, keep_b = state.ci_strs_a[ia], state.ci_strs_b[ib]
    wa = np.sum(np.abs(amps[ia]) ** 2, axis=1)
    wb = np.sum(np.abs(amps[:, ib]) ** 2, axis=0)
    if symmetrize_spin:
        both = _first_occurrences(_by_descending_count(np.concatenate((keep_a, keep_b)), np.concatenate((wa, wb))))
        return both, both
    return _by_descending_count(keep_a, wa), _by_descending_count(keep_b, wb)


# ---- SPMD plumbing (reference ``processes.py:100-134``: ``is_control_process`` / pickle ``broadcast`` over MPI;
# here the process group is torch.distributed's -- "nccl" = RCCL on MI355X, "gloo" on CPU)
def _process_group():
    import sys

    if "torch" not in sys.modules:  # a process group cannot exist without torch: no multi-second import for nothing
        return None
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover - torch is part of the image
        return None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _is_control_process(dist) -> bool:
    return dist is None or dist.get_rank() == 0


class _ControlError:
    """An exception raised on the control process, shipped in place of the broadcast's payload so that EVERY rank
    raises it (the reference leaves the other ranks blocked in the collective until their timeout)."""

    def __init__(self, exc: BaseException):
        self.kind, self.text = type(exc).__name__, str(exc)


def _broadcast(dist, obj):
    """Pickle broadcast from the control process (rank 0); a pass-through outside distributed mode."""
    if dist is None:
        if isinstance(obj, _ControlError):
            raise obj.exc
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=0)
    if isinstance(box[0], _ControlError):
        if isinstance(obj, _ControlError) and getattr(obj, "exc", None) is not None:
            raise obj.exc  # the control process re-raises the original exception
        raise RuntimeError(f"control process failed: {box[0].kind}: {box[0].text}")
    return box[0]


def _guarded(dist, fn):
    """Run a control-process section; an exception becomes a `_ControlError` payload for the broadcast that follows."""
    try:
        return fn()
    except Exception as exc:  # noqa: BLE001 - shipped to every rank by _broadcast
        err = _ControlError(exc)
        err.exc = exc
        return err


def diagonalize_fermionic_hamiltonian(
    one_body_tensor: np.ndarray,
    two_body_tensor: np.ndarray,
    bit_array,
    samples_per_batch: int,
    norb: int,
    nelec: tuple[int, int],
    *,
    num_batches: int = 1,
    energy_tol: float = 1e-8,
    occupancies_tol: float = 1e-5,
    max_iterations: int = 100,
    sci_solver: Callable[..., list[SCIResult]] | None = None,
    symmetrize_spin: bool = False,
    include_configurations=None,
    initial_occupancies: tuple[np.ndarray, np.ndarray] | None = None,
    carryover_threshold: float = 1e-4,
    max_dim: int | tuple[int, int] | None = None,
    callback: Callable[[list[SCIResult]], None] | None = None,
    seed: int | np.random.Generator | None = None,
) -> SCIResult:
    """Sample-based quantum diagonalization with self-consistent configuration recovery; returns the
    lowest-energy ``SCIResult`` seen.  Arguments as in the reference (``fermion.py:204-340``)."""
    if max_iterations < 1:
        raise ValueError("Maximum number of iterations must be at least 1.")
    n_alpha, n_beta = nelec
    if symmetrize_spin and n_alpha != n_beta:
        raise ValueError(
            "Spin symmetrization is only possible if the numbers of alpha and beta "
            f"electrons are equal. Instead, got {n_alpha} and {n_beta}."
        )
    max_dim_a, max_dim_b = max_dim if isinstance(max_dim, tuple) else (max_dim, max_dim)
    if symmetrize_spin and max_dim_a != max_dim_b:
        raise ValueError(
            "When requesting spin symmetrization, the maximum dimension must be "
            "the same for both spin alpha and spin beta. "
            f"Instead, got {max_dim_a} and {max_dim_b}"
        )
    if include_configurations is None:
        include_a = include_b = np.array([], dtype=int)
    elif isinstance(include_configurations, tuple):
        include_a, include_b = include_configurations
    else:
        include_a = include_b = include_configurations
    include_a, include_b = np.unique(include_a), np.unique(include_b)

    # Distributed (SPMD) mode, as in the reference (``fermion.py:410-451``): the control process (rank 0) owns the
    # random stream and everything that has no distributed implementation -- building the batches of CI strings,
    # picking the winner, convergence, carry-over -- and broadcasts the CI strings before and the iteration state
    # after the one collective step, ``sci_solver``.  The other ranks never draw random numbers, so ``seed=None``
    # is safe.  With a process group initialised the default solver is the batch-sharded collective one.
    dist = _process_group()
    control = _is_control_process(dist)
    rng = np.random.default_rng(seed) if control else None
    if sci_solver is not None:
        solver = sci_solver
    else:
        # the integrals are constant over the loop: read-only copies let this package's solvers find their device
        # context by identity instead of hashing 8 norb^4 bytes on every call (fermion._full_hash)
        one_body_tensor, two_body_tensor = freeze_integrals(one_body_tensor, two_body_tensor)
        if dist is not None:
            from .distributed import solve_sci_batch_distributed as solver
        else:
            solver = solve_sci_batch
    raw_bitstrings, raw_probs = bit_array_to_arrays(bit_array)

    occupancies = initial_occupancies
    best: SCIResult | None = None
    current: SCIResult | None = None
    carry_a = carry_b = np.array([], dtype=np.int64)

    for _ in range(max_iterations):
        # ---- configurations for this iteration (control process only)
        ci_strings = None

        def _prepare():
            if occupancies is None:
                bitstrings, probs = postselect_by_hamming_right_and_left(
                    raw_bitstrings, raw_probs, hamming_right=n_alpha, hamming_left=n_beta
                )
                if not bitstrings.size:
                    raise ValueError(
                        "The input bit array did not contain any valid bitstrings. "
                        "Either pass a bit array that contains at least one valid bitstring "
                        "(with the correct right and left Hamming weights), or specify a value for initial_occupancies."
                    )
            else:
                bitstrings, probs = recover_configurations(raw_bitstrings, raw_probs, occupancies, n_alpha, n_beta, rand_seed=rng)
            batches = subsample(bitstrings, probs, samples_per_batch=samples_per_batch, num_batches=num_batches, rand_seed=rng)
            return [
                _batch_strings(b, norb, symmetrize_spin, include_a, include_b, carry_a, carry_b, max_dim_a, max_dim_b)
                for b in batches
            ]

        if control:
            ci_strings = _guarded(dist, _prepare)
        ci_strings = _broadcast(dist, ci_strings)

        # ---- the seam (reference fermion.py:432): the only collective step
        results = solver(ci_strings, one_body_tensor, two_body_tensor, norb, nelec)

        # ---- bookkeeping (control process), then the iteration state to every rank (reference :436-451)
        state = None

        def _bookkeeping():
            if callback is not None:
                callback(results)
            winner = min(results, key=lambda r: r.energy)
            new_best = winner if best is None or winner.energy < best.energy else best
            converged = (
                current is not None
                and abs(current.energy - winner.energy) < energy_tol
                and np.linalg.norm(np.ravel(occupancies) - np.ravel(winner.orbital_occupancies), ord=np.inf) < occupancies_tol
            )
            carry = (carry_a, carry_b) if converged else _carryover(winner, carryover_threshold, symmetrize_spin)
            return (new_best, winner, bool(converged), carry)

        if control:
            state = _guarded(dist, _bookkeeping)
        best, winner, converged, (carry_a, carry_b) = _broadcast(dist, state)
        if converged:
            break
        current = winner
        occupancies = winner.orbital_occupancies

    return best
