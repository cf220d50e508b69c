"""The oracle is pinned before it is trusted (CPU, no GPU):
 * integer layer vs the golden vectors produced by the reference itself (tests/golden/make_golden.py)
   and vs the reference's literal known answers;
 * Slater-Condon construction (O1) vs the independent Jordan-Wigner brute force;
 * C restatement of pyscf's algorithm (O2, oracle/sci_ref.c) vs O1.
Floating-point layer: parity unpinned against pyscf (absent) -- see oracle/__init__.py."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import sqd_oracle as O

GOLD = json.loads((Path(__file__).parent / "golden" / "integer_layer.json").read_text())


@pytest.mark.parametrize("key", [k for k in GOLD if k.startswith("b2i_")])
def test_oracle_bitstring_matrix_to_integers_golden(key):
    case = GOLD[key]
    out = O.bitstring_matrix_to_integers(np.array(case["matrix"], dtype=bool))
    assert [str(int(x)) for x in out] == case["out"]
    assert str(out.dtype) == case["dtype"]


@pytest.mark.parametrize("key", [k for k in GOLD if k.startswith("ci_")])
def test_oracle_ci_strs_golden(key):
    case = GOLD[key]
    a, b = O.bitstring_matrix_to_ci_strs(np.array(case["matrix"], dtype=bool), open_shell=case["open_shell"])
    assert [str(int(x)) for x in a] == case["a"] and [str(int(x)) for x in b] == case["b"]


def test_oracle_ci_strs_reference_literals():
    # docs/guides/select_open_closed_shell.ipynb:180 and :438
    mat = np.array([[0, 0, 0, 1, 0, 0, 1, 0], [0, 1, 0, 0, 1, 0, 0, 0]], dtype=bool)
    a, b = O.bitstring_matrix_to_ci_strs(mat, open_shell=False)
    assert list(a) == [1, 2, 4, 8] and list(b) == [1, 2, 4, 8]
    a, b = O.bitstring_matrix_to_ci_strs(mat, open_shell=True)
    assert list(a) == [2, 8] and list(b) == [1, 4]
    # test/test_fermion.py:344-360: 57- and 64-bit strings survive the round trip
    for lit in ("1" * 20 + "0" * 37, "1" + "0" * 30 + "1" * 33):
        n = len(lit)
        row = np.array([c == "1" for c in lit + lit], dtype=bool)[None, :]
        a, _ = O.bitstring_matrix_to_ci_strs(row)
        assert format(int(a[0]), f"0{n}b") == lit


def test_oracle_check_ci_strs_golden():
    ok = GOLD["check_ok"]
    oa, ob = O.check_ci_strs((np.array(ok["a"]), np.array(ok["b"])))
    assert oa.tolist() == ok["out_a"] and ob.tolist() == ok["out_b"]
    for key in ("check_bad_up", "check_bad_dn"):
        case = GOLD[key]
        with pytest.raises(ValueError) as exc:
            O.check_ci_strs((np.array(case["a"]), np.array(case["b"])))
        assert str(exc.value) == case["error"]


@pytest.mark.parametrize("norb,nelec,na,nb,seed", [(5, (3, 2), 7, 6, 1), (5, (2, 3), 6, 8, 2), (6, (3, 3), 9, 9, 3), (4, (1, 2), 4, 5, 4)])
def test_slater_condon_matches_jordan_wigner(norb, nelec, na, nb, seed):
    h1, eri = O.synthetic_integrals(norb, seed=seed)
    sa = O.random_strings(norb, nelec[0], na, seed + 10)
    sb = O.random_strings(norb, nelec[1], nb, seed + 20)
    Hbf = O.jw_project(O.jw_hamiltonian(h1, eri), sa, sb, norb)
    assert np.allclose(Hbf, O.build_php(h1, eri, sa, sb, norb), atol=1e-12)
    assert np.allclose(np.diag(Hbf), O.make_hdiag(h1, eri, sa, sb, norb).ravel(), atol=1e-12)
    S2 = O.jw_project(O.jw_spin_square(norb), sa, sb, norb)
    assert np.allclose(S2, O.build_spin_square(sa, sb, norb, nelec), atol=1e-12)
    w, v = np.linalg.eigh(Hbf)
    c = v[:, 0].reshape(na, nb)
    d1a, d1b, d2 = O.jw_rdms(c, sa, sb, norb)
    r1a, r1b = O.make_rdm1s(c, sa, sb, norb)
    assert np.allclose(d1a, r1a, atol=1e-12) and np.allclose(d1b, r1b, atol=1e-12)
    assert np.allclose(d2, O.make_rdm2(c, sa, sb, norb), atol=1e-12)
    assert abs(O.energy_from_rdms(h1, eri, d1a + d1b, d2) - w[0]) < 1e-10
    aa, ab, bb = O.jw_rdm2s(c, sa, sb, norb)
    assert np.allclose(aa + bb + ab + ab.transpose(2, 3, 0, 1), d2, atol=1e-12)
    # the string-space sigma used at BASELINE sizes, against the dense matrix it factorises
    x = np.random.default_rng(seed).standard_normal((na, nb))
    assert np.allclose(O.sigma_string_space(h1, eri, sa, sb, x, norb).ravel(), Hbf @ x.ravel(), atol=1e-12)


def test_spin_complete_space_has_spin_eigenvalues():
    # full space (2e,3o): S^2 spectrum is {0, 2}
    norb = 3
    sa = sb = np.array([1, 2, 4])
    S2 = O.build_spin_square(sa, sb, norb, (1, 1))
    ev = np.round(np.linalg.eigvalsh(S2), 10)
    assert set(ev.tolist()) <= {0.0, 2.0}


def test_davidson_pyscf_flow_converges():
    norb, nelec = 6, (3, 3)
    h1, eri = O.synthetic_integrals(norb, seed=9)
    sa = O.hf_centred_strings(norb, 3, 15, 1)
    sb = O.hf_centred_strings(norb, 3, 14, 2)
    H = O.build_php(h1, eri, sa, sb, norb)
    hd = np.diag(H).copy()
    conv, e, x, ns = O.davidson_pyscf(lambda v: H @ v, O.init_guess(hd, 15, 14), hd)
    assert conv and abs(e - np.linalg.eigvalsh(H)[0]) < 1e-8


@pytest.mark.parametrize("norb,nelec,na,nb,seed", [(6, (3, 2), 12, 9, 5), (7, (3, 3), 20, 20, 7), (5, (1, 4), 5, 4, 9), (8, (4, 4), 30, 25, 2)])
def test_c_restatement_matches_numpy_oracle(norb, nelec, na, nb, seed):
    from oracle import sci_ref as R

    h1, eri = O.synthetic_integrals(norb, seed=seed)
    sa = O.random_strings(norb, nelec[0], na, seed + 1)
    sb = O.random_strings(norb, nelec[1], nb, seed + 2)
    H = O.build_php(h1, eri, sa, sb, norb)
    P = R.RefProblem(h1, eri, sa, sb)
    c = np.random.default_rng(0).standard_normal(na * nb)
    assert np.allclose(P.hdiag, np.diag(H), atol=1e-12)
    assert np.allclose(P.contract_2e(c), H @ c, atol=1e-11)
    # pyscf cre_des table: diagonal entries first, then exactly the oracle's in-set singles
    sl = O.single_links(sa, norb)
    cd = P.cd[0]
    nocc = nelec[0]
    assert (cd[:, :nocc, 3] == 1).all() and (cd[:, :nocc, 2] == np.arange(na)[:, None]).all()
    got = sorted((t, int(r[2]), int(r[0]), int(r[3])) for t in range(na) for r in cd[t, nocc:] if r[3] != 0)
    exp = sorted(zip(sl["tgt"].tolist(), sl["src"].tolist(), O.pair_index(sl["p"], sl["q"]).tolist(), sl["sign"].tolist()))
    assert got == exp
    e, amps, occ, nsig = R.solve_fermion_ref((sa, sb), h1, eri)
    assert abs(e - np.linalg.eigvalsh(H)[0]) < 1e-7
    r1a, r1b = O.make_rdm1s(amps, sa, sb, norb)
    assert np.allclose(occ[0], np.diag(r1a), atol=1e-12) and np.allclose(occ[1], np.diag(r1b), atol=1e-12)


@pytest.mark.parametrize("norb,nelec,na,nb,seed,hf", [(10, (5, 5), 40, 37, 11, True), (12, (4, 6), 30, 70, 13, False)])
def test_string_space_sigma_matches_dense_and_c_restatement(norb, nelec, na, nb, seed, hf):
    """O1s (sigma_string_space) == dense P H P == O2 (pyscf's gather/dgemm/scatter restatement)."""
    from oracle import sci_ref as R

    h1, eri = O.synthetic_integrals(norb, seed=seed)
    gen = O.hf_centred_strings if hf else O.random_strings
    sa, sb = gen(norb, nelec[0], na, seed + 1), gen(norb, nelec[1], nb, seed + 2)
    x = np.random.default_rng(seed).standard_normal((na, nb))
    ref = O.build_php(h1, eri, sa, sb, norb) @ x.ravel()
    assert np.allclose(O.sigma_string_space(h1, eri, sa, sb, x, norb).ravel(), ref, atol=1e-11)
    assert np.allclose(R.RefProblem(h1, eri, sa, sb).contract_2e(x), ref, atol=1e-11)


def test_init_guess_lower_triangle_rule():
    """pyscf direct_spin1._get_init_guess: the minimum is searched over A >= B when the spin sectors match."""
    h = np.array([[3.0, 1.0, 5.0], [2.0, 4.0, 6.0], [7.0, 8.0, 9.0]])  # global minimum at (0, 1), upper triangle
    x = O.init_guess(h.ravel(), 3, 3, nelec=(2, 2))
    assert np.argmax(x) == 1 * 3 + 0          # lowest element of the lower triangle: (1, 0) = 2.0
    x = O.init_guess(h.ravel(), 3, 3, nelec=(2, 1))
    assert np.argmax(x) == 0 * 3 + 1          # different sectors: the global minimum
    assert x[0] == 1e-5 and x[-1] == -1e-5
