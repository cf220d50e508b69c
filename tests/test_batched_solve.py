"""The batched native solve (``sqd_solve_batch``: tables, every Davidson round and the observables of ALL subspaces of a
``ci_strings`` list in the same kernel launches; reference seam ``qiskit_addon_sqd/fermion.py:432, :643-681``) against
the one-by-one solve of the same subspaces: every output bit for bit, and one anchor against the dense oracle.

CPU: through the kernel-logic emulator (tests/emu).  GPU: ``tests/test_gpu_parity.py`` holds the full-size cases."""
import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi, fermion
from qiskit_addon_sqd_amd.fermion import solve_sci, solve_sci_batch


def _batches(norb, nelec, sizes, hf):
    gen = O.hf_centred_strings if hf else O.random_strings
    return [(gen(norb, nelec[0], na, 11 + 3 * i), gen(norb, nelec[1], nb, 101 + 5 * i)) for i, (na, nb) in enumerate(sizes)]


def check_batched_equals_serial(h1, eri, norb, nelec, batches, spin_sq=None, **kw):
    serial = [solve_sci(b, h1, eri, norb, nelec, spin_sq=spin_sq, **kw) for b in batches]
    stats_serial = []
    for b in batches:  # (statistics of each one-by-one solve)
        solve_sci(b, h1, eri, norb, nelec, spin_sq=spin_sq, **kw)
        stats_serial.append(fermion.last_solve_stats())
    batched = solve_sci_batch(batches, h1, eri, norb, nelec, spin_sq=spin_sq, **kw)
    stats_b = fermion._TLS.batch_stats
    best = int(np.argmin([r.energy for r in serial]))
    for i, (s, r) in enumerate(zip(serial, batched)):
        assert r.energy == s.energy, (i, r.energy, s.energy)
        assert np.array_equal(r.orbital_occupancies[0], s.orbital_occupancies[0])
        assert np.array_equal(r.orbital_occupancies[1], s.orbital_occupancies[1])
        for k in ("converged", "iterations", "n_sigma", "e_davidson", "residual"):
            assert stats_b[i][k] == stats_serial[i][k], (i, k)
        # only the lowest-energy state came to the host with the call; the others are fetched when read
        assert (r.sci_state._pending_amplitudes() is not None) == (i != best)
        assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes), i
        assert np.array_equal(r.sci_state.ci_strs_a, s.sci_state.ci_strs_a)
    return serial, batched


def test_batched_solve_emulator(emu_backend):
    norb, nelec = 8, (4, 3)
    h1, eri = O.synthetic_integrals(norb, seed=5)
    # ragged sizes; connected sets take the work-item sigma kernel, the tiny ones the element-gather kernel: two launch
    # classes in one batch
    batches = _batches(norb, nelec, [(20, 16), (7, 5), (28, 12), (3, 2), (16, 20)], hf=True)
    serial, batched = check_batched_equals_serial(h1, eri, norb, nelec, batches)
    kinds = set()
    ctx = fermion._get_context(h1, eri, 0, slot="batch")
    assert ctx._batch_shapes == [(len(a), len(b)) for a, b in batches]
    # one oracle anchor: the dense eigensolver on the first subspace
    e_ref, *_ = O.solve_fermion_dense(batches[0], h1, eri)
    assert abs(batched[0].energy - e_ref) < 1e-8
    # the spin penalty (first form) goes through the batched launches too
    check_batched_equals_serial(h1, eri, norb, nelec, batches[:3], spin_sq=0.75)
    # RDMs on demand from a deferred state
    r = solve_sci_batch(batches[:2], h1, eri, norb, nelec)
    worst = int(np.argmax([x.energy for x in r]))
    assert abs(np.trace(r[worst].rdm1) - sum(nelec)) < 1e-9


def test_batched_solve_settles_deferred_states(emu_backend):
    """A result kept across the next batched solve still returns ITS state (fetched before the context is reused)."""
    norb, nelec = 6, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=2)
    b1 = _batches(norb, nelec, [(10, 9), (6, 7)], hf=True)
    b2 = _batches(norb, nelec, [(8, 8), (9, 5), (4, 4)], hf=False)
    first = solve_sci_batch(b1, h1, eri, norb, nelec)
    ref = [solve_sci(b, h1, eri, norb, nelec) for b in b1]
    ref2 = [solve_sci(b, h1, eri, norb, nelec) for b in b2]
    second = solve_sci_batch(b2, h1, eri, norb, nelec)
    # a slot keeps the latest and the previous call's solutions: nothing of `first` has been copied out yet ...
    lazy = [r for r in first if r.sci_state._pending_amplitudes() is not None]
    assert len(lazy) == 1 and lazy[0].sci_state.amplitudes.shape == lazy[0].sci_state.amplitudes.shape
    for r, s in zip(first, ref):
        assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
    # ... and a third call fetches what is still referenced of the second before its slots are reused
    third = solve_sci_batch(b1, h1, eri, norb, nelec)
    for r, s in zip(second, ref2):
        assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
    for r, s in zip(third, ref):
        assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
    assert len(second) == 3


def test_batched_solve_falls_back(emu_backend, monkeypatch):
    """Requests outside the batched path (a start vector, the per-round log, eager RDMs, one subspace) take the
    one-by-one path and still answer."""
    norb, nelec = 6, (3, 2)
    h1, eri = O.synthetic_integrals(norb, seed=3)
    batches = _batches(norb, nelec, [(10, 8), (5, 4)], hf=True)
    out = solve_sci_batch(batches, h1, eri, norb, nelec, compute_rdms=True)
    assert out[0].__dict__.get("rdm2") is not None
    one = solve_sci_batch(batches[:1], h1, eri, norb, nelec)
    assert isinstance(one[0].sci_state.amplitudes, np.ndarray)
    # the squared penalty form is solved one by one INSIDE the native batched call
    nel = (3, 3)
    b3 = _batches(norb, nel, [(8, 8), (6, 6)], hf=True)
    check_batched_equals_serial(h1, eri, norb, nel, b3, spin_sq=2.0)


@pytest.mark.gpu
@pytest.mark.parametrize("hf,nbatch", [(False, 8), (True, 16)])
def test_batched_solve_full_size_on_gpu(hip_lib, hf, nbatch):
    """BASELINE config 3's shape on one MI355X: 8 uniform / 16 HF-centred 317 x 317 subspaces of the N2-sized problem,
    batched against one by one -- energies, occupancies, Davidson statistics and states bit for bit."""
    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    gen = O.hf_centred_strings if hf else O.random_strings
    batches = [(gen(norb, 8, 317, 100 + i), gen(norb, 8, 317, 200 + i)) for i in range(nbatch)]
    serial, batched = check_batched_equals_serial(h1, eri, norb, nelec, batches)
    # an oracle anchor on one batch (string-space operator + pyscf-flow Davidson in numpy)
    k = nbatch - 1
    op = O.StringSpaceOperator(h1, eri, batches[k][0], batches[k][1], norb)
    hd = O.make_hdiag(h1, eri, batches[k][0], batches[k][1], norb).ravel()
    conv, e_ref, _, _ = O.davidson_pyscf(op, O.init_guess(hd, 317, 317, nelec), hd, tol=1e-12, max_cycle=200)
    assert conv and abs(batched[k].energy - e_ref) < 1e-8


@pytest.mark.gpu
def test_batched_solve_mixed_classes_on_gpu(hip_lib):
    """Ragged sizes and both sigma launch classes in one batch (element gather for the uniform sets, work items for the
    HF-centred ones, two template R of the latter), with and without the spin penalty, run twice (arena reuse)."""
    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    batches = [
        (O.random_strings(norb, 8, 317, 1), O.random_strings(norb, 8, 300, 2)),
        (O.hf_centred_strings(norb, 8, 200, 3), O.hf_centred_strings(norb, 8, 600, 4)),
        (O.hf_centred_strings(norb, 8, 317, 5), O.hf_centred_strings(norb, 8, 317, 6)),
        (O.random_strings(norb, 8, 50, 7), O.random_strings(norb, 8, 40, 8)),
        (O.hf_centred_strings(norb, 8, 90, 9), O.hf_centred_strings(norb, 8, 1100, 10)),
    ]
    for _ in range(2):
        check_batched_equals_serial(h1, eri, norb, nelec, batches)
    check_batched_equals_serial(h1, eri, norb, nelec, batches[:3], spin_sq=0.0)


@pytest.mark.gpu
def test_batched_solve_two_groups_on_gpu(hip_lib, monkeypatch):
    """SQD_BATCH_GROUPS=2 (experimental): the list cut into two interleaved groups, each a batched native solve on its own
    context and stream, from two host threads -- the same numbers as one by one, bit for bit."""
    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    batches = [(O.hf_centred_strings(norb, 8, 150 + 10 * i, 100 + i), O.hf_centred_strings(norb, 8, 140 + 7 * i, 200 + i)) for i in range(7)]
    serial = [solve_sci(b, h1, eri, norb, nelec) for b in batches]
    monkeypatch.setenv("SQD_BATCH_GROUPS", "2")
    for _ in range(2):
        grouped = solve_sci_batch(batches, h1, eri, norb, nelec)
        for s, r in zip(serial, grouped):
            assert r.energy == s.energy
            assert np.array_equal(r.orbital_occupancies[0], s.orbital_occupancies[0])
            assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)


def test_failed_batched_call_leaves_resident_states_readable(emu_backend):
    """A batched call that fails (here: a string list with mixed Hamming weights) must not move the bookkeeping of which
    call's states are resident: the deferred states of the call before it still resolve to the right amplitudes
    (the native slots rotated while the Python generation counter did not)."""
    norb, nelec = 6, (3, 2)
    h1, eri = O.synthetic_integrals(norb, seed=3)
    b1 = _batches(norb, nelec, [(10, 8), (7, 9), (5, 4)], hf=True)
    ref = [solve_sci(b, h1, eri, norb, nelec) for b in b1]
    first = solve_sci_batch(b1, h1, eri, norb, nelec)
    bad = [b1[0], (np.array([1, 3, 7]), b1[1][1]), b1[2]]  # popcounts 1, 2, 3
    with pytest.raises(ValueError):
        solve_sci_batch(bad, h1, eri, norb, nelec)
    for r, s in zip(first, ref):
        assert np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
    again = solve_sci_batch(b1, h1, eri, norb, nelec)
    for r, s in zip(again, ref):
        assert r.energy == s.energy and np.array_equal(r.sci_state.amplitudes, s.sci_state.amplitudes)
