"""Runs only where pyscf is importable (it is absent from the build container and from the GPU box, so these tests are
normally SKIPPED and the floating-point layer stays "parity unpinned against pyscf").  Where it
is present they pin the oracles -- and, on a GPU, the HIP path -- against the real thing, through exactly the calls the
reference makes (qiskit_addon_sqd/fermion.py:803-830): ``SelectedCI`` (+ ``fix_spin_``), ``kernel_fixed_space``,
``make_rdm1s`` / ``make_rdm1`` / ``make_rdm2``, ``spin_square``."""
import numpy as np
import pytest

pyscf = pytest.importorskip("pyscf")

from oracle import sqd_oracle as O  # noqa: E402

CASES = [(6, (3, 3), 14, 12, 5, None), (7, (4, 3), 20, 16, 9, None), (6, (3, 3), 14, 12, 5, 0.0), (7, (4, 3), 20, 16, 9, 0.75)]


def _pyscf_solve(h1, eri, sa, sb, norb, nelec, spin_sq, shift=0.1):
    from pyscf import fci

    myci = fci.selected_ci.SelectedCI()
    if spin_sq is not None:
        myci = fci.addons.fix_spin_(myci, ss=spin_sq, shift=shift)
    ci_strs = (np.asarray(sa, dtype=np.int64), np.asarray(sb, dtype=np.int64))
    _, vec = fci.selected_ci.kernel_fixed_space(myci, h1, eri, norb, nelec, ci_strs)
    dm1s = myci.make_rdm1s(vec, norb, nelec)
    dm1 = myci.make_rdm1(vec, norb, nelec)
    dm2 = myci.make_rdm2(vec, norb, nelec)
    e = np.einsum("pr,pr->", dm1, h1) + 0.5 * np.einsum("prqs,prqs->", dm2, eri)
    return e, np.array(vec), (np.diagonal(dm1s[0]), np.diagonal(dm1s[1])), myci.spin_square(vec, norb, nelec)[0], dm1, dm2


def _problem(norb, nelec, na, nb, seed):
    h1, eri = O.synthetic_integrals(norb, seed=seed)
    return h1, eri, O.random_strings(norb, nelec[0], na, seed + 1), O.random_strings(norb, nelec[1], nb, seed + 2)


@pytest.mark.parametrize("norb,nelec,na,nb,seed,spin_sq", CASES)
def test_oracle_against_pyscf(norb, nelec, na, nb, seed, spin_sq):
    h1, eri, sa, sb = _problem(norb, nelec, na, nb, seed)
    e, vec, occ, s2, dm1, dm2 = _pyscf_solve(h1, eri, sa, sb, norb, nelec, spin_sq)
    # operator level: pyscf's contract_2e (through its own absorb_h1e) against the Slater-Condon oracle
    from pyscf import fci

    myci = fci.selected_ci.SelectedCI()
    x = np.random.default_rng(seed).standard_normal((na, nb))
    civ = fci.selected_ci._as_SCIvector(x, (np.asarray(sa, dtype=np.int64), np.asarray(sb, dtype=np.int64)))
    h2e = myci.absorb_h1e(h1, eri, norb, nelec, 0.5)
    sig = np.asarray(myci.contract_2e(h2e, civ, norb, nelec))
    assert np.abs(sig - O.sigma_string_space(h1, eri, sa, sb, x, norb)).max() < 1e-10
    assert np.abs(np.asarray(myci.make_hdiag(h1, eri, (np.asarray(sa), np.asarray(sb)), norb, nelec)).reshape(na, nb)
                  - O.make_hdiag(h1, eri, sa, sb, norb)).max() < 1e-10
    # solution level: the oracle's dense solve of the same (penalised) problem
    e_o, _state_o, occ_o, s2_o, _ = O.solve_fermion_dense((sa, sb), h1, eri, spin_sq=spin_sq)
    assert abs(e - e_o) < 1e-6 and abs(s2 - s2_o) < 1e-5
    assert np.allclose(occ[0], occ_o[0], atol=1e-5) and np.allclose(occ[1], occ_o[1], atol=1e-5)
    assert np.allclose(dm1, sum(O.make_rdm1s(vec, sa, sb, norb)), atol=1e-10)
    assert np.allclose(dm2, O.make_rdm2(vec, sa, sb, norb), atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("norb,nelec,na,nb,seed,spin_sq", CASES)
def test_hip_path_against_pyscf(norb, nelec, na, nb, seed, spin_sq):
    from qiskit_addon_sqd_amd.fermion import solve_fermion

    h1, eri, sa, sb = _problem(norb, nelec, na, nb, seed)
    e, vec, occ, s2, _, _ = _pyscf_solve(h1, eri, sa, sb, norb, nelec, spin_sq)
    e_g, state, occ_g, s2_g = solve_fermion((sa, sb), h1, eri, spin_sq=spin_sq)
    assert abs(e_g - e) < 1e-6 and abs(s2_g - s2) < 1e-5  # north_star bar: 1e-6 Ha
    assert np.allclose(occ_g[0], occ[0], atol=1e-5) and np.allclose(occ_g[1], occ[1], atol=1e-5)
    assert abs(abs(np.vdot(state.amplitudes, vec)) - 1.0) < 1e-6
