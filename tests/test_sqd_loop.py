"""The SQD outer loop and its sample-processing helpers against the REFERENCE's behaviour.

``tests/golden/sqd_loop.json`` was produced by running the reference's own
``diagonalize_fermionic_hamiltonian`` (under import stubs, tests/golden/make_golden.py) with a
deterministic solver plug-in built on the numpy oracle, recording every list of CI strings that
crossed the ``sci_solver`` seam.  Replaying the same inputs through this package's loop with the same
plug-in must reproduce those lists bit for bit (same numpy Generator stream: post-selection,
configuration recovery, subsampling, ordering, truncation, carry-over) and the same final result.
Literal known answers are from the reference's tests (cited inline)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd.fermion import SCIResult, SCIState
from qiskit_addon_sqd_amd.sampling import (bit_array_to_arrays, counts_to_arrays, postselect_by_hamming_right_and_left,
                                           recover_configurations, subsample)
from qiskit_addon_sqd_amd.sqd import diagonalize_fermionic_hamiltonian

GOLD = json.loads((Path(__file__).parent / "golden" / "sqd_loop.json").read_text())


def _oracle_solver(record):
    def solver(ci_strings, one, two, norb, nelec):
        record.append([(np.asarray(a).copy(), np.asarray(b).copy()) for a, b in ci_strings])
        out = []
        for sa, sb in ci_strings:
            e, amps, occ, _, _ = O.solve_fermion_dense((sa, sb), one, two)
            out.append(SCIResult(e, SCIState(amps, sa, sb, norb, nelec), occ))
        return out

    return solver


@pytest.mark.parametrize("name", sorted(GOLD))
def test_loop_reproduces_reference_seam(name):
    g = GOLD[name]
    norb, nelec = g["norb"], tuple(g["nelec"])
    h1, eri = O.synthetic_integrals(norb, seed=g["integrals_seed"])
    calls = []
    res = diagonalize_fermionic_hamiltonian(
        h1, eri, np.array(g["noisy"], dtype=bool), samples_per_batch=g["samples_per_batch"], norb=norb, nelec=nelec,
        num_batches=g["num_batches"], max_iterations=g["max_iterations"], symmetrize_spin=g["symmetrize_spin"],
        max_dim=g["max_dim"], sci_solver=_oracle_solver(calls), carryover_threshold=g["carryover_threshold"],
        seed=g["seed"])
    assert len(calls) == len(g["calls"])
    for it, (mine, ref) in enumerate(zip(calls, g["calls"])):
        assert len(mine) == len(ref)
        for (a, b), r in zip(mine, ref):
            assert a.tolist() == r["a"] and b.tolist() == r["b"], f"iteration {it}"
            assert str(a.dtype) == r["dtype"]
    assert res.energy == pytest.approx(g["energy"], abs=1e-12)
    assert res.sci_state.ci_strs_a.tolist() == g["strs_a"] and res.sci_state.ci_strs_b.tolist() == g["strs_b"]
    assert np.allclose(np.abs(res.sci_state.amplitudes), np.array(g["abs_amplitudes"]), atol=1e-12)
    assert np.allclose(res.orbital_occupancies[0], g["occ_a"], atol=1e-12)


def test_postselect_known_answer():
    # reference test/test_subsampling.py:53-74
    mat = np.array([[1, 0, 1, 1, 0, 0, 1, 1], [1, 1, 1, 1, 0, 0, 1, 1], [0, 1, 1, 1, 1, 1, 0, 0], [1, 0, 0, 0, 0, 0, 1, 1]],
                   dtype=bool)
    probs = np.array([0.1, 0.2, 0.4, 0.3])
    rows = [i for i in range(4) if mat[i, 4:].sum() == 2 and mat[i, :4].sum() == 3]
    out, p = postselect_by_hamming_right_and_left(mat, probs.copy(), hamming_right=2, hamming_left=3)
    assert np.array_equal(out, mat[rows]) and np.allclose(p, probs[rows] / probs[rows].sum())
    with pytest.raises(ValueError, match="non-negative"):
        postselect_by_hamming_right_and_left(mat, probs, hamming_right=-1, hamming_left=1)
    with pytest.raises(ValueError, match="must be even"):
        postselect_by_hamming_right_and_left(mat[:, :7], probs, hamming_right=1, hamming_left=1)


def test_recover_configurations_deterministic_cases():
    # reference test/test_configuration_recovery.py:70-108: occupancies force the outcome
    zeros = np.zeros((1, 4), dtype=bool)
    out, p = recover_configurations(zeros, [1.0], (np.array([1.0, 1.0]), np.array([1.0, 1.0])), 2, 2, rand_seed=4224)
    assert out.tolist() == [[True] * 4] and p.tolist() == [1.0]
    ones = np.ones((1, 4), dtype=bool)
    out, _ = recover_configurations(ones, [1.0], (np.array([0.0, 0.0]), np.array([0.0, 0.0])), 0, 0, rand_seed=4224)
    assert out.tolist() == [[False] * 4]
    # duplicates after recovery are merged and probabilities renormalised
    mat = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [1, 1, 1, 1]], dtype=bool)
    out, p = recover_configurations(mat, [0.25, 0.25, 0.5], (np.array([0.9, 0.1]), np.array([0.9, 0.1])), 1, 1, rand_seed=7)
    assert out.shape[1] == 4 and abs(p.sum() - 1) < 1e-12 and (out[:, :2].sum(1) == 1).all() and (out[:, 2:].sum(1) == 1).all()
    with pytest.raises(ValueError, match="non-negative"):
        recover_configurations(mat, [0.25, 0.25, 0.5], (np.zeros(2), np.zeros(2)), -1, 1)


def test_subsample_shapes_and_errors():
    rng = np.random.default_rng(0)
    mat = rng.integers(2, size=(50, 6)).astype(bool)
    probs = np.full(50, 1 / 50)
    b = subsample(mat, probs, 10, 3, rand_seed=1)
    assert len(b) == 3 and all(x.shape == (10, 6) for x in b)
    assert all(np.array_equal(x, mat) for x in subsample(mat, probs, 60, 2, rand_seed=1))
    assert len(subsample(np.empty((0, 6), dtype=bool), np.array([]), 5, 2)) == 2
    with pytest.raises(ValueError, match="Samples per batch"):
        subsample(mat, probs, 0, 1)
    with pytest.raises(ValueError, match="number of batches"):
        subsample(mat, probs, 5, 0)


def test_bit_array_duck_typing_and_counts():
    class FakeBitArray:  # the attributes of qiskit.primitives.BitArray that the reference reads
        def __init__(self, bools):
            pad = (-bools.shape[1]) % 8
            self.array = np.packbits(np.concatenate([np.zeros((bools.shape[0], pad), bool), bools], 1), -1)
            self.num_bits, self.num_shots = bools.shape[1], bools.shape[0]

    bools = np.array([[1, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]], dtype=bool)
    m1, p1 = bit_array_to_arrays(FakeBitArray(bools))
    m2, p2 = bit_array_to_arrays(bools)
    assert np.array_equal(m1, m2) and np.allclose(p1, p2) and abs(p1.sum() - 1) < 1e-15
    m, p = counts_to_arrays({"101": 2, "011": 1, "111": 1})
    assert m.shape == (3, 3) and np.allclose(p, [0.5, 0.25, 0.25])


def test_loop_argument_validation():
    h1, eri = O.synthetic_integrals(4, seed=1)
    bits = np.zeros((4, 8), dtype=bool)
    with pytest.raises(ValueError, match="at least 1"):
        diagonalize_fermionic_hamiltonian(h1, eri, bits, 2, 4, (2, 2), max_iterations=0)
    with pytest.raises(ValueError, match="Spin symmetrization"):
        diagonalize_fermionic_hamiltonian(h1, eri, bits, 2, 4, (2, 1), symmetrize_spin=True)
    with pytest.raises(ValueError, match="maximum dimension"):
        diagonalize_fermionic_hamiltonian(h1, eri, bits, 2, 4, (2, 2), symmetrize_spin=True, max_dim=(2, 3))
    with pytest.raises(ValueError, match="did not contain any valid bitstrings"):
        diagonalize_fermionic_hamiltonian(h1, eri, bits, 2, 4, (2, 2), sci_solver=_oracle_solver([]))


def test_recover_configurations_native_replay_matches_numpy_stream():
    """The native row repair (csrc/sqd_recover.hip) replays numpy's Generator.choice(p=, replace=False): same
    repaired rows, same merged probabilities, and the generator left at the same position as the per-row
    Python path -- on noisy samples with several wrong bits per half, zero weights and duplicate rows."""
    from qiskit_addon_sqd_amd import sampling

    rng0 = np.random.default_rng(123)
    norb, na, nb = 11, 4, 3
    for trial in range(6):
        n = 400
        bits = np.zeros((n, 2 * norb), dtype=bool)
        for i in range(n):
            bits[i, rng0.choice(norb, nb, replace=False)] = True
            bits[i, norb + rng0.choice(norb, na, replace=False)] = True
        bits ^= rng0.random(bits.shape) < (0.05 + 0.05 * trial)
        bits[::7] = bits[0]  # duplicates
        probs = rng0.random(n)
        probs /= probs.sum()
        occ = (rng0.random(norb), rng0.random(norb))
        if trial % 2:
            occ[0][:3] = 0.0  # exact zeros / ones in the occupancies: zero flip weights
            occ[1][-2:] = 1.0
        g_fast, g_slow = np.random.default_rng(900 + trial), np.random.default_rng(900 + trial)
        m_fast, p_fast = sampling.recover_configurations(bits, probs, occ, na, nb, g_fast)
        orig = sampling._recover_rows_native, sampling._recover_all_native
        sampling._recover_rows_native = lambda *a, **k: False  # force the per-row numpy path
        sampling._recover_all_native = lambda *a, **k: None
        try:
            m_slow, p_slow = sampling.recover_configurations(bits, probs, occ, na, nb, g_slow)
        finally:
            sampling._recover_rows_native, sampling._recover_all_native = orig
        assert np.array_equal(m_fast, m_slow) and np.array_equal(p_fast, p_slow)
        assert g_fast.random() == g_slow.random()  # same stream position
        assert (m_fast[:, :norb].sum(axis=1) == nb).all() and (m_fast[:, norb:].sum(axis=1) == na).all()
    # a generator that cannot be rewound falls back to the per-row path
    mt = np.random.Generator(np.random.MT19937(5))
    m, p = sampling.recover_configurations(bits, probs, occ, na, nb, mt)
    assert (m[:, norb:].sum(axis=1) == na).all()


def test_unique_rows_matches_numpy_axis0():
    from qiskit_addon_sqd_amd.sampling import _unique_rows

    rng = np.random.default_rng(3)
    for nbits in (1, 7, 8, 9, 60, 64, 65, 130):
        b = rng.random((500, nbits)) < 0.5
        b[::5] = b[1]
        r0, c0 = np.unique(b, axis=0, return_counts=True)
        r1, c1 = _unique_rows(b)
        assert np.array_equal(r0, r1) and np.array_equal(c0, c1), nbits


def test_native_sample_processing_matches_numpy():
    """The native helpers of the loop's host side (libsqd_hip.so, no GPU needed: `sqd_choice_replay`, `sqd_hamming_excess`
    + `sqd_recover_rows` on all rows + `sqd_merge_rows`) against the numpy code paths they replace: same indices, same
    rows, same probabilities bit for bit, and the SAME position of the random stream afterwards."""
    from qiskit_addon_sqd_amd import sampling as SP

    rng0 = np.random.default_rng(5)
    # --- subsampling draws: many collisions (size close to n) and few (size << n)
    for n, size, nb in ((12, 9, 5), (400, 37, 8), (5000, 250, 3)):
        p = rng0.random(n)
        p[rng0.random(n) < 0.2] = 0.0
        if np.count_nonzero(p) < size:
            p[:size] = 0.5
        p /= p.sum()
        a, b = np.random.default_rng(99), np.random.default_rng(99)
        got = SP._choice_native(a, p, size, nb)
        assert got is not None
        ref = [b.choice(np.arange(n), size, replace=False, p=p) for _ in range(nb)]
        assert all(np.array_equal(g, r) for g, r in zip(got, ref))
        assert a.bit_generator.state == b.bit_generator.state
    # --- inputs numpy raises on are left to numpy (and the stream is not moved)
    a = np.random.default_rng(1)
    s0 = a.bit_generator.state
    assert SP._choice_native(a, np.array([0.5, 0.5, 0.0]), 3, 1) is None and a.bit_generator.state == s0
    # --- whole recover_configurations: native three-pass path against the numpy path on the same input and stream
    norb, n = 9, 3000
    bits = rng0.random((n, 2 * norb)) < 0.4
    bits = np.concatenate([bits, bits[:200]])  # duplicates
    probs = rng0.random(len(bits))
    probs /= probs.sum()
    occ = (rng0.random(norb), rng0.random(norb))
    a, b = np.random.default_rng(7), np.random.default_rng(7)
    a.integers(0, 10, dtype=np.uint32), b.integers(0, 10, dtype=np.uint32)  # leaves a cached 32-bit half in the generator
    assert a.bit_generator.state["has_uint32"] == 1
    m1, f1 = SP.recover_configurations(bits, probs, occ, 4, 3, rand_seed=a)
    real = SP._recover_all_native
    SP._recover_all_native = lambda *args, **kw: None  # force the numpy path
    try:
        m2, f2 = SP.recover_configurations(bits, probs, occ, 4, 3, rand_seed=b)
    finally:
        SP._recover_all_native = real
    assert np.array_equal(m1, m2) and np.array_equal(f1, f2)
    assert a.bit_generator.state == b.bit_generator.state


def test_native_repair_without_divisions_equals_the_replay_and_numpy():
    """`sqd_recover_rows` locates every draw among running sums of the raw flip weights (no divisions) and falls back to
    the operation-by-operation replay of numpy only near a tie.  At the loop's own shape (30 orbitals, 2e4 noisy samples,
    several wrong bits per half, exact zeros among the weights): rows and stream consumption equal the replay's
    (``SQD_RECOVER_EXACT`` in a fresh process) and, on a slice, the per-row numpy path's."""
    import subprocess
    import sys

    code = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from qiskit_addon_sqd_amd import sampling
rng0 = np.random.default_rng(77)
norb, na, nb, n = 30, 8, 7, 20000
bits = np.zeros((n, 2 * norb), dtype=bool)
cols = np.argsort(rng0.random((n, norb)), axis=1)
np.put_along_axis(bits[:, :norb], cols[:, :nb], True, axis=1)
cols = np.argsort(rng0.random((n, norb)), axis=1)
np.put_along_axis(bits[:, norb:], cols[:, :na], True, axis=1)
bits ^= rng0.random(bits.shape) < 0.04
bits[::9] = bits[3]
probs = rng0.random(n); probs /= probs.sum()
occ = (rng0.random(norb), rng0.random(norb))
occ[0][:4] = 0.0; occ[1][-3:] = 1.0; occ[1][5] = occ[1][6]  # zero weights, equal weights
g = np.random.default_rng(5)
m, p = sampling.recover_configurations(bits, probs, occ, na, nb, g)
print(hashlib.sha256(m.tobytes()).hexdigest(), hashlib.sha256(p.tobytes()).hexdigest(), g.random(), len(m))
g = np.random.default_rng(6)
m, p = sampling.recover_configurations(bits[:1500], probs[:1500] / probs[:1500].sum(), occ, na, nb, g)
print(hashlib.sha256(m.tobytes()).hexdigest(), hashlib.sha256(p.tobytes()).hexdigest(), g.random(), len(m))
""" % str(Path(__file__).resolve().parent.parent)
    import os

    outs = []
    for exact in (False, True):
        env = dict(os.environ)
        env.pop("SQD_RECOVER_EXACT", None)
        if exact:
            env["SQD_RECOVER_EXACT"] = "1"
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 2
    # the slice against the per-row numpy path, in this process
    import hashlib

    from qiskit_addon_sqd_amd import sampling

    rng0 = np.random.default_rng(77)
    norb, na, nb, n = 30, 8, 7, 20000
    bits = np.zeros((n, 2 * norb), dtype=bool)
    cols = np.argsort(rng0.random((n, norb)), axis=1)
    np.put_along_axis(bits[:, :norb], cols[:, :nb], True, axis=1)
    cols = np.argsort(rng0.random((n, norb)), axis=1)
    np.put_along_axis(bits[:, norb:], cols[:, :na], True, axis=1)
    bits ^= rng0.random(bits.shape) < 0.04
    bits[::9] = bits[3]
    probs = rng0.random(n)
    probs /= probs.sum()
    occ = (rng0.random(norb), rng0.random(norb))
    occ[0][:4] = 0.0
    occ[1][-3:] = 1.0
    occ[1][5] = occ[1][6]
    orig = sampling._recover_rows_native, sampling._recover_all_native
    sampling._recover_rows_native = lambda *a, **k: False
    sampling._recover_all_native = lambda *a, **k: None
    try:
        g = np.random.default_rng(6)
        m, p = sampling.recover_configurations(bits[:1500], probs[:1500] / probs[:1500].sum(), occ, na, nb, g)
    finally:
        sampling._recover_rows_native, sampling._recover_all_native = orig
    line = f"{hashlib.sha256(m.tobytes()).hexdigest()} {hashlib.sha256(p.tobytes()).hexdigest()} {g.random()} {len(m)}"
    assert outs[0].splitlines()[1] == line


def test_batch_strings_union_shortcut_and_carryover_mask():
    """Without ``max_dim`` the ordered merge of `_batch_strings` reduces to a sorted union (the order only decides what a
    truncation drops); `_carryover` finds the rows / columns holding a large amplitude from one mask.  Both against the
    formulations they replace (reference ``fermion.py:520-553``, ``:607-631``)."""
    from qiskit_addon_sqd_amd import sqd as L

    rng = np.random.default_rng(2)
    norb = 9
    for sym in (False, True):
        for trial in range(4):
            samples = rng.random((60, 2 * norb)) < 0.4
            inc_a, inc_b = np.unique(rng.integers(0, 1 << norb, 3)), np.unique(rng.integers(0, 1 << norb, 2))
            car_a, car_b = rng.permutation(1 << norb)[:7].astype(np.int64), rng.permutation(1 << norb)[:5].astype(np.int64)
            if sym:
                car_b = car_a
            fast = L._batch_strings(samples, norb, sym, inc_a, inc_b, car_a, car_b, None, None)
            slow = L._batch_strings(samples, norb, sym, inc_a, inc_b, car_a, car_b, 10**9, 10**9)
            assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1])
    for sym in (False, True):
        na, nb = 23, 17
        amps = rng.standard_normal((na, nb)) * np.exp(-rng.random((na, nb)) * 3) * np.outer(0.5 ** np.arange(na), 0.6 ** np.arange(nb))
        amps /= np.linalg.norm(amps)
        sa, sb = np.sort(rng.permutation(1 << norb)[:na]), np.sort(rng.permutation(1 << norb)[:nb])
        res = SCIResult(-1.0, SCIState(amps, sa, sb, norb, (3, 3)), (np.zeros(norb), np.zeros(norb)))
        thr = 1e-3
        got = L._carryover(res, thr, sym)
        mag = np.abs(amps.reshape(-1))
        order = np.argsort(mag)
        big = order[np.searchsorted(mag, thr, sorter=order):]
        ia, ib = np.divmod(big, nb)
        ia, ib = np.unique(ia), np.unique(ib)
        wa, wb = np.sum(np.abs(amps[ia]) ** 2, axis=1), np.sum(np.abs(amps[:, ib]) ** 2, axis=0)
        if sym:
            both = np.concatenate((sa[ia], sb[ib]))[np.argsort(np.concatenate((wa, wb)))[::-1]]
            _, idx = np.unique(both, return_index=True)
            want = (both[np.sort(idx)],) * 2
        else:
            want = (sa[ia][np.argsort(wa)[::-1]], sb[ib][np.argsort(wb)[::-1]])
        assert 0 < len(ia) < na and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
