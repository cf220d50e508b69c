"""Parity tests proper (``-m gpu``): the HIP path, called through the C ABI of libsqd_hip.so on a
real MI355X, against the numpy oracle on the same seeded inputs; plus size-independent properties at
the BASELINE sizes where the dense oracle cannot go.

Tolerances: CI-string addressing (targets, sources, orbitals, pair indices, signs) bit-exact;
sigma / hdiag <= 1e-11 * |H|max * sqrt(D) absolute; energies <= 1e-8 Ha against dense ``eigh``
(north_star bar: 1e-6 Ha); RDMs 1e-12.
"""
import os
from pathlib import Path

import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi

ROOT = Path(__file__).resolve().parents[1]

from _parity import check_link_tables, check_operators, make_problem, run_full_parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "norb,nelec,na,nb,seed,hf",
    [
        (6, (3, 2), 12, 9, 5, False),
        (7, (3, 3), 20, 20, 7, True),
        (5, (1, 4), 5, 4, 9, False),
        (4, (2, 2), 6, 6, 3, False),      # FCI limit
        (10, (5, 5), 40, 37, 11, True),   # D = 1480
        (12, (4, 6), 30, 70, 13, False),  # ragged: nb crosses a 64-slice boundary
        (8, (4, 4), 70, 70, 17, False),   # complete alpha/beta spaces C(8,4)=70 -> FCI; 36 doubles/string > ELL cap
        (12, (2, 6), 3, 924, 19, False),  # complete beta space: 36 singles + 225 doubles per string (overflow rows)
    ],
)
def test_full_parity(hip_lib, norb, nelec, na, nb, seed, hf):
    run_full_parity(hip_lib, norb, nelec, na, nb, seed, hf, with_rdm2=(norb <= 10))


@pytest.mark.parametrize("direct", ["0", "1"])
def test_direct_and_work_item_sigma_forced(hip_lib, monkeypatch, direct):
    """Both sigma kernels on the same inputs: SQD_SIGMA_DIRECT=1 forces the element-gather kernel (the default only
    for ultra-sparse string sets such as the uniform headline batch), 0 forbids it."""
    monkeypatch.setenv("SQD_SIGMA_DIRECT", direct)
    run_full_parity(hip_lib, 7, (3, 3), 20, 20, 7, True)
    run_full_parity(hip_lib, 12, (4, 6), 30, 70, 13, False, with_rdm2=False)
    # the headline size, uniform strings, against the string-space oracle
    norb, nelec, h1, eri, sa, sb = _n2_problem(317, False)
    x = np.random.default_rng(5).standard_normal((317, 317))
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        s = ctx.sigma(x)
        p = ctx.sigma(x, 1, 0.0, 0.3)
        ss = ctx.contract_ss(x)
    assert np.abs(s - O.sigma_string_space(h1, eri, sa, sb, x, norb)).max() < 1e-10
    S2x = O.build_spin_square(sa, sb, norb, nelec, sparse=True) @ x.ravel()
    assert np.abs(ss.ravel() - S2x).max() < 1e-11
    assert np.abs(p.ravel() - (s.ravel() + 0.3 * S2x)).max() < 1e-10


@pytest.mark.parametrize("rows", ["1", "2", "3", "6", "8"])
def test_rows_sigma_kernel_forced(hip_lib, monkeypatch, rows):
    """k_sigma_rows (R whole rows of C per workgroup in LDS; the default for large uniform-random sets) forced at
    sizes the oracles check: all operator forms, Davidson, HF-centred and uniform sets, the headline size."""
    monkeypatch.setenv("SQD_SIGMA_ROWS", rows)
    run_full_parity(hip_lib, 7, (3, 3), 20, 20, 7, True)
    run_full_parity(hip_lib, 12, (4, 6), 30, 70, 13, False, with_rdm2=False)
    for hf in (False, True):
        norb, nelec, h1, eri, sa, sb = _n2_problem(317, hf)
        x = np.random.default_rng(5).standard_normal((317, 317))
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == f"k_sigma_rows<{rows}>"
            s = ctx.sigma(x)
            p = ctx.sigma(x, 1, 0.0, 0.3)
            ss = ctx.contract_ss(x)
        ref = O.sigma_string_space(h1, eri, sa, sb, x, norb)
        assert np.abs(s - ref).max() < 1e-11 * max(1.0, np.abs(ref).max())
        S2x = O.build_spin_square(sa, sb, norb, nelec, sparse=True) @ x.ravel()
        assert np.abs(ss.ravel() - S2x).max() < 1e-11 * max(1.0, np.abs(S2x).max())
        assert np.abs(p.ravel() - (s.ravel() + 0.3 * S2x)).max() < 1e-10 * max(1.0, np.abs(ref).max())


def test_rows_sigma_kernel_default_selection(hip_lib, monkeypatch):
    """Uniform 2048 x 2048 (D = 4.2e6, ~4 links per string): the rows kernel is the default; its sigma, S^2 and
    penalty forms agree with the work-item kernel forced on the same inputs, and a whole solve agrees in energy."""
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, 2048, 21), S.uniform_strings(30, 8, 2048, 22)
    x = np.random.default_rng(6).standard_normal((2048, 2048))
    out = {}
    for forced in (None, "0"):
        if forced is None:
            monkeypatch.delenv("SQD_SIGMA_ROWS", raising=False)
        else:
            monkeypatch.setenv("SQD_SIGMA_ROWS", forced)
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            kern = ctx.sigma_kernel()
            assert kern.startswith("k_sigma_rows") if forced is None else kern == "k_sigma"
            amps, st = ctx.davidson()
            out[forced] = (ctx.sigma(x), ctx.contract_ss(x), ctx.sigma(x, 1, 0.0, 0.3), st["e_davidson"], st["converged"])
    a, b = out[None], out["0"]
    scale = np.abs(b[0]).max()
    # the default-selected rows kernel against the ORACLE (O1s: string-space operator in numpy, ~20 s at this size),
    # not only against the product's other kernel
    hd_max = np.abs(O.make_hdiag(h1, eri, sa, sb, 30)).max()
    assert np.abs(a[0] - O.sigma_string_space(h1, eri, sa, sb, x, 30)).max() < 1e-11 * hd_max
    assert np.abs(a[0] - b[0]).max() < 1e-12 * scale
    assert np.abs(a[1] - b[1]).max() < 1e-12 * np.abs(b[1]).max()
    assert np.abs(a[2] - b[2]).max() < 1e-12 * scale
    assert a[4] == 1 and b[4] == 1 and abs(a[3] - b[3]) < 1e-9


@pytest.mark.parametrize("na,nb,rows", [(1500, 1111, "3"), (1111, 2050, "6"), (700, 4097, "1")])
def test_rows_sigma_kernel_ragged_against_work_items(hip_lib, monkeypatch, na, nb, rows):
    """Ragged shapes (last workgroup short, last 64-column slice partial, rectangular), every operator form incl. the
    squared spin penalty: k_sigma_rows forced against the work-item kernel forced, same inputs."""
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, na, 31), S.uniform_strings(30, 7, nb, 32)
    x = np.random.default_rng(8).standard_normal((na, nb))
    out = {}
    for forced in (rows, "0"):
        monkeypatch.setenv("SQD_SIGMA_ROWS", forced)
        monkeypatch.setenv("SQD_SIGMA_DIRECT", "0") if forced == "0" else monkeypatch.delenv("SQD_SIGMA_DIRECT", raising=False)
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == (f"k_sigma_rows<{rows}>" if forced != "0" else "k_sigma")
            out[forced] = (ctx.sigma(x), ctx.contract_ss(x), ctx.sigma(x, 1, 0.75, 0.3), ctx.sigma(x, 2, 0.75, 0.3))
    for a, b in zip(out[rows], out["0"]):
        assert np.abs(a - b).max() < 1e-12 * max(1.0, np.abs(b).max())


def test_long_rows_kernel_selection_and_oracle(hip_lib):
    """256 x 10 000 uniform strings (the 1e4-column regime of BASELINE config 2 read literally, at a row count the
    oracle finishes in seconds): the default selection must be k_sigma_rows<2>, its sigma must match the oracle (O1s)
    and be Hermitian on a pair of vectors."""
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, 256, 41), S.uniform_strings(30, 8, 10000, 42)
    rng = np.random.default_rng(9)
    x, y = rng.standard_normal((256, 10000)), rng.standard_normal((256, 10000))
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_sigma_rows<2>"
        sx, sy = ctx.sigma(x), ctx.sigma(y)
    hd_max = np.abs(O.make_hdiag(h1, eri, sa, sb, 30)).max()
    assert np.abs(sx - O.sigma_string_space(h1, eri, sa, sb, x, 30)).max() < 1e-11 * hd_max
    assert abs(np.vdot(y, sx) - np.vdot(sy, x)) < 1e-9 * abs(np.vdot(y, sx))


def test_capped_ell_overflow_rows(hip_lib, monkeypatch):
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    run_full_parity(hip_lib, 7, (3, 3), 20, 20, 7, True)
    run_full_parity(hip_lib, 10, (5, 5), 40, 37, 11, True)


def test_many_axpy_items_forced(hip_lib, monkeypatch):
    # many small AXPY items per row: partial rows + the fixed-order reduce with many slots per row
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_L", "2")
    run_full_parity(hip_lib, 10, (5, 5), 40, 37, 11, True)
    run_full_parity(hip_lib, 8, (4, 4), 70, 70, 17, False, variants=False)


def test_global_row_fallback_forced(hip_lib, monkeypatch):
    # the path for rows that do not fit LDS, forced at a size the oracle can check (3 column chunks)
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_GLOBAL_ROWS", "64")
    run_full_parity(hip_lib, 10, (5, 5), 40, 150, 31, True)
    run_full_parity(hip_lib, 12, (2, 6), 3, 924, 19, False)


def test_multi_pass_partial_sums_forced(hip_lib, monkeypatch):
    # LDS room for 48 partial sums per list: the staged row's virtual rows are walked in many passes
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_PASS", "48")
    run_full_parity(hip_lib, 10, (5, 5), 40, 150, 31, True)
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    run_full_parity(hip_lib, 12, (2, 6), 3, 924, 19, False)


def test_h2_sto3g(hip_lib):
    h1 = np.diag([-1.2525, -0.4759])
    eri = np.zeros((2, 2, 2, 2))
    eri[0, 0, 0, 0], eri[1, 1, 1, 1] = 0.6746, 0.6974
    eri[0, 0, 1, 1] = eri[1, 1, 0, 0] = 0.6636
    for p, q, r, s in [(0, 1, 0, 1), (0, 1, 1, 0), (1, 0, 0, 1), (1, 0, 1, 0)]:
        eri[p, q, r, s] = 0.1813
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace([1, 2], [1, 2])
        _, st = ctx.davidson()
        e = ctx.energy()
    H = O.jw_project(O.jw_hamiltonian(h1, eri), [1, 2], [1, 2], 2)
    assert abs(e - np.linalg.eigvalsh(H)[0]) < 1e-10
    assert abs(e + 0.7137 + 1.1373) < 1e-3


def test_wide_orbitals_addressing(hip_lib):
    # norb = 64: strings use bit 63 (uint64), pair indices up to 2079
    norb, nelec = 64, (3, 2)
    rng = np.random.default_rng(3)
    h1 = rng.standard_normal((norb, norb)); h1 = 0.5 * (h1 + h1.T)
    # cheap 8-fold symmetric eri: rank-2 density fitting
    B = rng.standard_normal((2, norb, norb)) * 0.1; B = 0.5 * (B + B.transpose(0, 2, 1))
    eri = np.einsum("Lpq,Lrs->pqrs", B, B)
    def strings(ne, n, seed):
        r = np.random.default_rng(seed); out = {int(sum(1 << int(p) for p in (63, 62, 0)[:ne]))}
        while len(out) < n:
            out.add(int(sum(1 << int(p) for p in r.choice(norb, ne, replace=False))))
        return np.array(sorted(out), dtype=np.uint64)
    sa, sb = strings(3, 40, 1), strings(2, 33, 2)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        check_link_tables(ctx, sa, sb, norb, h1, eri)
        H = O.build_php(h1, eri, sa, sb, norb)
        c = rng.standard_normal((len(sa), len(sb)))
        assert np.allclose(ctx.sigma(c).ravel(), H @ c.ravel(), atol=1e-10)


def test_invalid_inputs(hip_lib):
    h1, eri, sa, sb = make_problem(6, (3, 2), 5, 4, 1)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        bad = sa.copy(); bad[2] = 0b1111
        with pytest.raises(ValueError, match="hamming weight"):
            ctx.set_subspace(np.sort(bad), sb)
        with pytest.raises(ValueError, match="strictly ascending"):
            ctx.set_subspace(sa[::-1].copy(), sb)
        with pytest.raises(ValueError, match="empty"):
            ctx.set_subspace(np.array([], dtype=np.int64), sb)


def _n2_problem(n, hf, seed=0):
    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    gen = O.hf_centred_strings if hf else O.random_strings
    return norb, nelec, h1, eri, gen(norb, 8, n, seed + 1), gen(norb, 8, n, seed + 2)


@pytest.mark.parametrize("hf", [False, True])
def test_n2_headline_properties(hip_lib, hf):
    """N2 (16e,30o), 317 x 317 = 100 489 determinants (the metric's size): properties that do not need
    a dense oracle -- hermiticity, linearity, link-table bit-exactness at full size, diagonal,
    variational bound, <S^2>, trace of dm1, energy from RDMs == <c|H|c>."""
    norb, nelec, h1, eri, sa, sb = _n2_problem(317, hf)
    rng = np.random.default_rng(1)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        check_link_tables(ctx, sa, sb, norb, h1, eri)
        assert np.allclose(ctx.hdiag(), O.make_hdiag(h1, eri, sa, sb, norb), atol=1e-10)
        x = rng.standard_normal((317, 317)); y = rng.standard_normal((317, 317))
        sx, sy = ctx.sigma(x), ctx.sigma(y)
        assert abs(np.vdot(y, sx) - np.vdot(x, sy)) < 1e-8 * abs(np.vdot(y, sx))
        assert np.allclose(ctx.sigma(2.0 * x - 3.0 * y), 2.0 * sx - 3.0 * sy, atol=1e-9)
        amps, st = ctx.davidson()
        assert st["converged"] == 1
        e = ctx.energy()
        assert abs(e - st["e_davidson"]) < 1e-8
        assert e <= ctx.hdiag().min() + 1e-9          # variational: below the best determinant
        r = ctx.sigma(amps) - e * amps                # eigen-residual
        assert np.linalg.norm(r) < 1e-4
        d1a, d1b = ctx.rdm1s()
        assert abs(np.trace(d1a) - 8) < 1e-9 and abs(np.trace(d1b) - 8) < 1e-9
        assert np.allclose(d1a, d1a.T, atol=1e-10)
        r1a, r1b = O.make_rdm1s(amps, sa, sb, norb)
        assert np.allclose(d1a, r1a, atol=1e-11) and np.allclose(d1b, r1b, atol=1e-11)
        d2 = ctx.rdm2()
        assert abs(O.energy_from_rdms(h1, eri, d1a + d1b, d2) - e) < 1e-8
        assert abs(np.einsum("ppqq->", d2) - 16 * 15) < 1e-7
        s2 = ctx.spin_square()
        assert s2 > -1e-9


def _full_size_checks(hip_lib, norb, nocc, n, hf, seeds, with_o2, e_tol=1e-8, kernel=None):
    """sigma, hdiag, E0, occupancies and the state itself at a BASELINE size against the oracles:
    O1s = string-space evaluation of the decomposition that ``build_php`` forms densely (numpy, independent of
    the J-table / hdiag split of the kernels), O2 = the C restatement of pyscf's contract_2e.  Tolerances:
    sigma / hdiag 1e-11 * max|hdiag| absolute (observed ~1e-13); E0 1e-8 Ha (north_star bar 1e-6 Ha)."""
    from oracle import sci_ref as R

    h1, eri = O.synthetic_integrals(norb)
    gen = O.hf_centred_strings if hf else O.random_strings
    sa, sb = gen(norb, nocc, n, seeds[0]), gen(norb, nocc, n, seeds[1])
    x = np.random.default_rng(7).standard_normal((n, n))
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        if kernel is not None:
            assert ctx.sigma_kernel() == kernel
        hd = ctx.hdiag()
        s_gpu = ctx.sigma(x)
        x0 = ctx.init_guess()
        # (occupancies and the state are compared at 5e-6 / 1e-9 below: first order in the residual, so the run uses
        # the tight residual rule; the default rule -- pyscf's sqrt(tol) without a penalty -- is checked on the energy)
        _, st_def = ctx.davidson(fetch=False)
        e_def = st_def["e_davidson"]
        amps, st = ctx.davidson(tol_residual=np.sqrt(1e-9) / 32.0)
        e, s2, occ_a, occ_b = ctx.observables()
        assert st_def["converged"] == 1 and st_def["n_sigma"] <= st["n_sigma"] and abs(e_def - e) < 1e-8
    scale = np.abs(hd).max()
    assert np.allclose(hd, O.make_hdiag(h1, eri, sa, sb, norb), rtol=0, atol=1e-11 * scale)
    assert np.abs(s_gpu - O.sigma_string_space(h1, eri, sa, sb, x, norb)).max() < 1e-11 * scale
    if with_o2:
        prob = R.RefProblem(h1, eri, sa, sb)
        assert np.abs(s_gpu.ravel() - prob.contract_2e(x)).max() < 1e-11 * scale
        assert np.allclose(hd.ravel(), prob.hdiag, rtol=0, atol=1e-11 * scale)
    # start vector = pyscf get_init_guess
    g = O.init_guess(hd.ravel(), n, n, nelec=(nocc, nocc))
    assert np.allclose(x0.ravel(), g / np.linalg.norm(g), rtol=0, atol=1e-15)
    # ground state: the oracle's Davidson (pyscf control flow, tightened so that its own residual is below the
    # comparison tolerances) on the string-space operator
    op = O.StringSpaceOperator(h1, eri, sa, sb, norb)
    conv, e_ref, x_ref, _ = O.davidson_pyscf(op, g, hd.ravel(), tol=1e-13, max_cycle=200)
    assert conv and st["converged"] == 1
    x_ref = x_ref / np.linalg.norm(x_ref)
    assert abs(e - e_ref) < e_tol and abs(st["e_davidson"] - e_ref) < e_tol
    assert abs(abs(np.vdot(amps.ravel(), x_ref)) - 1.0) < 1e-9
    r1a, r1b = O.make_rdm1s(x_ref.reshape(n, n), sa, sb, norb)
    assert np.allclose(occ_a, np.diag(r1a), atol=5e-6) and np.allclose(occ_b, np.diag(r1b), atol=5e-6)
    return e, e_ref


@pytest.mark.parametrize("hf,dense", [(False, None), (True, "1"), (True, "0")])
def test_n2_full_size_against_oracles(hip_lib, monkeypatch, hf, dense):
    """BASELINE headline size, N2 (16e,30o) 317 x 317 = 100 489 determinants, both string generators: the full
    sigma vector against O1s and O2, E0 / occupancies / state against the oracle's own Davidson.  The HF-centred set
    (same-spin blocks 22-26 % dense) selects the matrix-core same-spin product by default; it is run in that mode and
    with the sparse same-spin work items (SQD_SIGMA_DENSE=0)."""
    kernel = "k_sigma_direct"
    if hf:
        if dense == "0":
            monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
        kernel = "k_same_spin_mfma+k_sigma" if dense == "1" else "k_sigma"
    _full_size_checks(hip_lib, 30, 8, 317, hf, (1, 2), with_o2=True, kernel=kernel)


@pytest.mark.parametrize("hf,dense", [(False, None), (True, "1"), (True, "0")])
def test_fes_full_size_against_oracle(hip_lib, monkeypatch, hf, dense):
    """BASELINE config 4's size, (30e,40o) 707 x 707 = 499 849 determinants: full sigma, E0, occupancies against
    O1s (O2's dense formulation needs 1.3e14 flop per sigma at this size -- minutes -- and is left out)."""
    if dense == "0":
        monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
    _full_size_checks(hip_lib, 40, 15, 707, hf, (5, 6), with_o2=False)


def test_n2_uniform_solve_vs_reference_flow(hip_lib):
    """The whole reference orchestration (O2: check strings -> pyscf-flow Davidson on the restated contract_2e
    -> <c|H|c>, occupancies; ``oracle/sci_ref.py: solve_fermion_ref`` follows fermion.py:745-845) at the headline
    size, fed with a BOOL bitstring matrix as the reference's users do (fermion.py:788-795)."""
    from oracle import sci_ref as R
    from qiskit_addon_sqd_amd.fermion import bitstring_matrix_to_ci_strs, solve_fermion

    norb, nelec, h1, eri, sa, sb = _n2_problem(317, False)
    mat = O.bitstring_matrix_from_strings(sa, sb, norb)   # 317 samples |b_i a_i>
    assert mat.dtype == bool and mat.shape == (317, 60)
    ci = bitstring_matrix_to_ci_strs(mat, open_shell=True)
    assert np.array_equal(ci[0], sa) and np.array_equal(ci[1], sb)
    e, state, occ, s2 = solve_fermion(mat, h1, eri, open_shell=True)
    e_ref, amps_ref, occ_ref, _ = R.solve_fermion_ref(O.bitstring_matrix_to_ci_strs(mat, open_shell=True), h1, eri,
                                                      tol=1e-12)
    assert abs(e - e_ref) < 1e-8
    assert abs(abs(np.vdot(state.amplitudes, amps_ref)) - 1.0) < 1e-8
    assert np.allclose(occ[0], occ_ref[0], atol=5e-6) and np.allclose(occ[1], occ_ref[1], atol=5e-6)
    # closed shell: both spins get the union (fermion.py:1032-1033)
    e2, state2, _, _ = solve_fermion(mat[:40], h1, eri, open_shell=False)
    u = np.union1d(sa[:40], sb[:40])
    assert np.array_equal(state2.ci_strs_a, u) and np.array_equal(state2.ci_strs_b, u)
    assert state2.amplitudes.shape == (len(u), len(u))


def test_config3_eight_batches_one_gpu(hip_lib):
    """BASELINE config 3 on one GPU: 8 independent 317 x 317 subsample batches of the N2-sized problem through
    (i) ``solve_sci_batch`` -- the batched native solve and the threads-x-streams path -- and (ii) the collective ``solve_sci_batch_distributed``
    on an RCCL ("nccl") process group of world size 1 -- all-reduce of the (E, occ) records and winner broadcast
    included -- against the one-at-a-time run, bit for bit."""
    import socket

    import torch
    import torch.distributed as dist

    from qiskit_addon_sqd_amd.distributed import solve_sci_batch_distributed
    from qiskit_addon_sqd_amd.fermion import solve_sci_batch

    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    batches = [(O.random_strings(norb, 8, 317, 100 + i), O.random_strings(norb, 8, 317, 200 + i)) for i in range(7)]
    batches.append((O.hf_centred_strings(norb, 8, 317, 1), O.hf_centred_strings(norb, 8, 317, 2)))
    serial = solve_sci_batch(batches, h1, eri, norb, nelec, compute_rdms=False, concurrency=1)
    par = solve_sci_batch(batches, h1, eri, norb, nelec, compute_rdms=False)  # (the batched native solve)
    thr = solve_sci_batch(batches, h1, eri, norb, nelec, compute_rdms=False, concurrency=4)  # (threads x streams)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        coll = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False)
        mean = solve_sci_batch_distributed(batches, h1, eri, norb, nelec, compute_rdms=False, occupancy_reduce="mean")
    finally:
        dist.destroy_process_group()
    best = int(np.argmin([r.energy for r in serial]))
    assert best == 7  # the HF-centred batch contains the aufbau determinant
    for i, (s0, p, t, c) in enumerate(zip(serial, par, thr, coll)):
        for r in (p, t, c):
            assert r.energy == s0.energy, i
            assert np.array_equal(r.orbital_occupancies[0], s0.orbital_occupancies[0])
            assert np.array_equal(r.orbital_occupancies[1], s0.orbital_occupancies[1])
            assert np.array_equal(r.sci_state.amplitudes, s0.sci_state.amplitudes)
    mean_a = np.mean([s0.orbital_occupancies[0] for s0 in serial], axis=0)
    assert all(np.allclose(r.orbital_occupancies[0], mean_a, atol=1e-13) for r in mean)
    # one oracle anchor for the batch set: the winner's energy
    op = O.StringSpaceOperator(h1, eri, batches[7][0], batches[7][1], norb)
    hd = O.make_hdiag(h1, eri, batches[7][0], batches[7][1], norb).ravel()
    conv, e_ref, _, _ = O.davidson_pyscf(op, O.init_guess(hd, 317, 317, nelec), hd, tol=1e-12, max_cycle=200)
    assert conv and abs(serial[7].energy - e_ref) < 1e-8


def test_fes_size_properties(hip_lib):
    """BASELINE config 4's size: 40 orbitals, 15 + 15 electrons, 707 x 707 = 499 849 determinants (HF-centred
    strings).  Size-independent properties only: link tables bit-exact at full size, diagonal, hermiticity and
    linearity of sigma, the variational bound, the eigen-residual, traces of the RDMs, energy from the RDMs."""
    norb, nocc, n = 40, 15, 707
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, nocc, n, 5), O.hf_centred_strings(norb, nocc, n, 6)
    rng = np.random.default_rng(3)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        check_link_tables(ctx, sa, sb, norb, h1, eri)
        assert np.allclose(ctx.hdiag(), O.make_hdiag(h1, eri, sa, sb, norb), atol=1e-9)
        x = rng.standard_normal((n, n)); y = rng.standard_normal((n, n))
        sx, sy = ctx.sigma(x), ctx.sigma(y)
        assert abs(np.vdot(y, sx) - np.vdot(x, sy)) < 1e-8 * abs(np.vdot(y, sx))
        assert np.allclose(ctx.sigma(2.0 * x - 3.0 * y), 2.0 * sx - 3.0 * sy, atol=1e-8)
        amps, st = ctx.davidson()
        assert st["converged"] == 1
        e = ctx.energy()
        assert abs(e - st["e_davidson"]) < 1e-8
        assert e <= ctx.hdiag().min() + 1e-9
        assert np.linalg.norm(ctx.sigma(amps) - e * amps) < 1e-4
        d1a, d1b = ctx.rdm1s()
        assert abs(np.trace(d1a) - nocc) < 1e-9 and abs(np.trace(d1b) - nocc) < 1e-9
        d2 = ctx.rdm2()
        assert abs(O.energy_from_rdms(h1, eri, d1a + d1b, d2) - e) < 1e-8
        assert abs(np.einsum("ppqq->", d2) - 30 * 29) < 1e-6
        assert ctx.spin_square() > -1e-9


def test_n2_sigma_vs_sparse_oracle(hip_lib):
    """sigma at N2 size against the scipy-sparse Slater-Condon oracle on a 60 x 50 sub-selection
    (D = 3000) drawn from the headline HF-centred string sets."""
    norb, nelec, h1, eri, sa, sb = _n2_problem(317, True)
    sa, sb = sa[:60], sb[:50]
    rng = np.random.default_rng(2)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        H, S2 = check_operators(ctx, h1, eri, sa, sb, norb, nelec, rng)
        amps, st = ctx.davidson()
        w = np.linalg.eigvalsh(H)
        assert abs(ctx.energy() - w[0]) < 1e-8


@pytest.mark.parametrize("hf", [False, True])
@pytest.mark.parametrize("nmany", [10000, 20000])
def test_long_rows_against_transposed_problem(hip_lib, hf, nmany):
    """20 000 beta strings: a C row (160 KB) does not fit LDS, so sigma takes the global-row / column-chunk
    path; 10 000: the row (80 KB) is staged but one partial sum per virtual row is not, so the beta lists
    are walked in passes.  With spin-restricted integrals the problem is symmetric under alpha<->beta
    exchange (a global sign on the basis), so sigma of the transposed problem -- nmany alpha strings x 6
    beta strings, the plain LDS-staged path -- must give the transposed result; contract_ss and the penalty
    form likewise."""
    norb = 30
    h1, eri = O.synthetic_integrals(norb)
    gen = O.hf_centred_strings if hf else O.random_strings
    few, many = gen(norb, 8, 6, 41), gen(norb, 8, nmany, 43)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, nmany))
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(few, many)
        s_long = ctx.sigma(x)
        p_long = ctx.sigma(x, use_spin=1, ss=0.0, shift=0.3)
        ss_long = ctx.contract_ss(x)
        ctx.set_subspace(many, few)
        xt = np.ascontiguousarray(x.T)
        s_t = ctx.sigma(xt)
        p_t = ctx.sigma(xt, use_spin=1, ss=0.0, shift=0.3)
        ss_t = ctx.contract_ss(xt)
    scale = np.abs(s_t).max()
    assert np.abs(s_long - s_t.T).max() < 1e-11 * scale
    assert np.abs(p_long - p_t.T).max() < 1e-11 * scale
    assert np.abs(ss_long - ss_t.T).max() < 1e-11 * np.abs(ss_t).max()


def test_variational_monotonicity(hip_lib):
    """Enlarging the subspace can only lower E0."""
    norb, nelec, h1, eri, sa, sb = _n2_problem(200, True)
    es = []
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        for n in (50, 100, 200):
            ctx.set_subspace(sa[:n], sb[:n])   # prefixes of sorted sets are subsets
            ctx.davidson()
            es.append(ctx.energy())
    assert es[0] >= es[1] - 1e-9 >= es[2] - 2e-9


def test_python_api_solve_fermion(hip_lib):
    from qiskit_addon_sqd_amd.fermion import SCIState, solve_fermion, solve_sci, solve_sci_batch

    norb, nelec = 8, (4, 3)
    h1, eri = O.synthetic_integrals(norb)
    sa = O.hf_centred_strings(norb, 4, 20, 1)
    sb = O.hf_centred_strings(norb, 3, 16, 2)
    for spin_sq in (None, 0.75, 3.75):
        e, state, occ, s2 = solve_fermion((sa, sb), h1, eri, spin_sq=spin_sq)
        e_ref, amps_ref, occ_ref, s2_ref, _ = O.solve_fermion_dense((sa, sb), h1, eri, spin_sq=spin_sq)
        # <c|H|c> of a penalty-shifted eigenvector is first order in the Davidson residual (1e-6):
        # 5e-7 Ha here, inside the north_star bar of 1e-6 Ha
        assert abs(e - e_ref) < 5e-7, spin_sq
        # occupancies and <S^2> are first order in the Davidson residual: sqrt(tol)/32 = 1e-6 with a penalty, pyscf's
        # sqrt(tol) = 3e-5 without (the default rule follows the reference's solver)
        otol = 1e-5 if spin_sq is not None else 3e-4
        assert np.allclose(occ[0], occ_ref[0], atol=otol) and np.allclose(occ[1], occ_ref[1], atol=otol)
        assert abs(s2 - s2_ref) < otol
        assert isinstance(state, SCIState) and state.amplitudes.shape == (20, 16)
        assert abs(state.spin_square() - s2) < 1e-9
    res = solve_sci((sa, sb), h1, eri, norb, nelec, spin_sq=0.75)
    H = O.build_php(h1, eri, sa, sb, norb); S2 = O.build_spin_square(sa, sb, norb, nelec)
    w, v = np.linalg.eigh(H + 0.2 * (S2 - 0.75 * np.eye(len(H))))
    assert abs(res.energy - v[:, 0] @ H @ v[:, 0]) < 5e-7  # penalised state: first order in |r| (see above)
    assert res.rdm1.shape == (8, 8) and res.rdm2.shape == (8, 8, 8, 8)
    assert np.allclose(res.sci_state.rdm(1, spin_summed=True), res.rdm1, atol=1e-12)
    batch = solve_sci_batch([(sa, sb), (sa[:10], sb[:8])], h1, eri, norb, nelec)
    assert len(batch) == 2 and batch[0].energy <= batch[1].energy + 1e-9


def test_sqd_loop_end_to_end_on_gpu(hip_lib):
    """Whole SQD loop with the HIP solver as ``sci_solver`` against the same loop driven by the dense
    numpy oracle: identical CI strings at the seam every iteration, energies within 1e-8 Ha."""
    import json
    from pathlib import Path

    from qiskit_addon_sqd_amd.fermion import SCIResult, SCIState, solve_sci_batch
    from qiskit_addon_sqd_amd.sqd import diagonalize_fermionic_hamiltonian

    g = json.loads((Path(__file__).parent / "golden" / "sqd_loop.json").read_text())["loop_open"]
    norb, nelec = g["norb"], tuple(g["nelec"])
    h1, eri = O.synthetic_integrals(norb, seed=g["integrals_seed"])
    seen = []

    def gpu_solver(ci_strings, one, two, norb_, nelec_):
        seen.append([(np.asarray(a).copy(), np.asarray(b).copy()) for a, b in ci_strings])
        return solve_sci_batch(ci_strings, one, two, norb_, nelec_, tol_residual=1e-9)

    res = diagonalize_fermionic_hamiltonian(
        h1, eri, np.array(g["noisy"], dtype=bool), samples_per_batch=g["samples_per_batch"], norb=norb, nelec=nelec,
        num_batches=g["num_batches"], max_iterations=g["max_iterations"], symmetrize_spin=False, max_dim=None,
        sci_solver=gpu_solver, carryover_threshold=g["carryover_threshold"], seed=g["seed"])
    assert len(seen) == len(g["calls"])
    for mine, ref in zip(seen, g["calls"]):
        for (a, b), r in zip(mine, ref):
            assert a.tolist() == r["a"] and b.tolist() == r["b"]
    assert abs(res.energy - g["energy"]) < 1e-8
    assert np.allclose(np.abs(res.sci_state.amplitudes), np.array(g["abs_amplitudes"]), atol=1e-6)


def test_concurrent_batches_on_one_gpu(hip_lib):
    """``concurrency=k``: k host threads, each with its own context + stream on the same device; results
    must equal the one-at-a-time run bit for bit (every reduction is fixed-order)."""
    from qiskit_addon_sqd_amd.fermion import solve_sci_batch

    norb, nelec = 10, (5, 5)
    h1, eri = O.synthetic_integrals(norb)
    batches = [(O.hf_centred_strings(norb, 5, 30 + 3 * i, 10 + i), O.hf_centred_strings(norb, 5, 28 + 2 * i, 40 + i))
               for i in range(6)]
    seq = solve_sci_batch(batches, h1, eri, norb, nelec, spin_sq=0.0, compute_rdms=False)
    par = solve_sci_batch(batches, h1, eri, norb, nelec, spin_sq=0.0, compute_rdms=False, concurrency=3)
    for a, b in zip(seq, par):
        assert a.energy == b.energy
        assert np.array_equal(a.sci_state.amplitudes, b.sci_state.amplitudes)
        assert np.array_equal(a.orbital_occupancies[0], b.orbital_occupancies[0])


def test_context_on_caller_stream(hip_lib):
    """sqd_ctx_use_stream: the same solve on a torch-owned stream gives the same bits (fixed-order reductions),
    and the caller's stream survives the context."""
    import torch

    stream = torch.cuda.Stream()
    h1, eri, sa, sb = make_problem(8, (4, 4), 30, 30, 3, True)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        a0, _ = ctx.davidson()
        e0 = ctx.energy()
        ctx.use_stream(stream.cuda_stream)
        ctx.set_subspace(sa, sb)
        a1, _ = ctx.davidson()
        e1 = ctx.energy()
    assert e0 == e1 and np.array_equal(a0, a1)
    with torch.cuda.stream(stream):  # still usable after the context is gone
        x = torch.ones(8, device="cuda").sum()
    stream.synchronize()
    assert float(x) == 8.0


def test_row_sharded_sigma_on_gpu(hip_lib, monkeypatch):
    """SURVEY 8f-3 on the real kernels: a context that owns alpha rows [row0, row1) of the N2-sized 317 x 317 problem
    must reproduce the ORACLE's sigma rows (string-space operator, numpy) and exactly those rows of the whole-subspace
    sigma of the same kernels (same work items, same order: bit for bit), for H, for the spin-penalised operator and
    for S^2; and the collective solver on an RCCL group of world size 1 must agree with the single-GPU solver."""
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (a row shard runs the sparse same-spin work items: compare like with like)
    import socket

    import torch
    import torch.distributed as dist

    from qiskit_addon_sqd_amd.fermion import solve_sci
    from qiskit_addon_sqd_amd.sharded import solve_sci_sharded

    norb, nelec, h1, eri, sa, sb = _n2_problem(317, True)
    x = np.random.default_rng(11).standard_normal((317, 317))
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        s_full, p_full, ss_full, hd_full = ctx.sigma(x), ctx.sigma(x, 1, 0.0, 0.25), ctx.contract_ss(x), ctx.hdiag()
        s_oracle = O.sigma_string_space(h1, eri, sa, sb, x, norb)  # the oracle, not the product's own whole-subspace sigma
        scale = np.abs(hd_full).max()
        xd = torch.from_numpy(x).cuda()
        for lo, hi in ((0, 317), (100, 250), (316, 317)):
            ctx.set_subspace_rows(sa, sb, lo, hi)
            out = torch.empty((hi - lo, 317), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            ctx.sigma_rows_dev(xd.data_ptr(), out.data_ptr())
            ctx.sync()
            assert np.abs(out.cpu().numpy() - s_oracle[lo:hi]).max() < 1e-11 * scale
            assert np.array_equal(out.cpu().numpy(), s_full[lo:hi])
            ctx.sigma_rows_dev(xd.data_ptr(), out.data_ptr(), 1, 0.0, 0.25)
            ctx.sync()
            assert np.array_equal(out.cpu().numpy(), p_full[lo:hi])
            ctx.contract_ss_rows_dev(xd.data_ptr(), out.data_ptr())
            ctx.sync()
            assert np.array_equal(out.cpu().numpy(), ss_full[lo:hi])
            assert np.array_equal(ctx.hdiag(), hd_full[lo:hi])
            if (lo, hi) != (0, 317):  # whole-vector entry points refuse a shard
                with pytest.raises(_capi.SQDNativeError, match="row shard"):
                    ctx.sigma(x)
    ref = solve_sci((sa, sb), h1, eri, norb, nelec, compute_rdms=False)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        res = solve_sci_sharded((sa, sb), h1, eri, norb, nelec)  # a group of one: ONE native call per iteration
        monkeypatch.setenv("SQD_SHARD_FORCE_COLLECTIVES", "1")    # ... against the staged calls around the collectives
        staged = solve_sci_sharded((sa, sb), h1, eri, norb, nelec)
        monkeypatch.delenv("SQD_SHARD_FORCE_COLLECTIVES")
    finally:
        dist.destroy_process_group()
    assert staged._sharded_stats["converged"] and staged.energy == res.energy
    assert np.array_equal(staged.sci_state.amplitudes, res.sci_state.amplitudes)
    assert res._sharded_stats["converged"]
    assert abs(res.energy - ref.energy) < 1e-8
    assert abs(abs(np.vdot(res.sci_state.amplitudes, ref.sci_state.amplitudes)) - 1.0) < 1e-8
    assert np.allclose(res.orbital_occupancies[0], ref.orbital_occupancies[0], atol=5e-6)


def test_row_sharded_overlap_hf_1000_on_gpu(hip_lib, monkeypatch):
    """The collective row-sharded solver on a CONNECTED subspace, HF-centred 1000 x 1000 (D = 1e6), with its collectives
    really issued on an RCCL group of one (SQD_SHARD_FORCE_COLLECTIVES): the sigma stage runs as two native calls around
    the asynchronous all-gather (own-row work items on the send buffer while the gather is in flight, the rest behind
    it).  Same energy and state as the one-call stage (bit for bit) and as the single-GPU solver (1e-8)."""
    import socket

    import torch
    import torch.distributed as dist

    from qiskit_addon_sqd_amd.fermion import solve_sci
    from qiskit_addon_sqd_amd.sharded import solve_sci_sharded

    norb, nelec = 30, (8, 8)
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, 1000, 11), O.hf_centred_strings(norb, 8, 1000, 13)
    ref = solve_sci((sa, sb), h1, eri, norb, nelec, compute_rdms=False)
    monkeypatch.setenv("SQD_SHARD_FORCE_COLLECTIVES", "1")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        out = {}
        # "items": the sparse work items, the kernel a real row shard runs and the one whose own-row items run in front of
        # the gather; "default": a group of one holds all rows and takes the sparse-product path, which runs whole
        # behind the gather (the two-call protocol must still be right)
        for kern in ("items", "default"):
            if kern == "items":
                monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
                monkeypatch.setenv("SQD_SIGMA_SPMM", "0")
            else:
                monkeypatch.delenv("SQD_SIGMA_DENSE")
                monkeypatch.delenv("SQD_SIGMA_SPMM")
            for ov in ("1", "0"):
                monkeypatch.setenv("SQD_SHARD_OVERLAP", ov)
                out[kern, ov] = solve_sci_sharded((sa, sb), h1, eri, norb, nelec)
        # the linear spin penalty on the default path (the whole-row kernel's SPIN form; its split rows in the sharded dots)
        pen = solve_sci_sharded((sa, sb), h1, eri, norb, nelec, spin_sq=0.0)
    finally:
        dist.destroy_process_group()
    monkeypatch.delenv("SQD_SHARD_FORCE_COLLECTIVES")
    pen_ref = solve_sci((sa, sb), h1, eri, norb, nelec, spin_sq=0.0, compute_rdms=False)
    assert pen._sharded_stats["converged"] and abs(pen.energy - pen_ref.energy) < 1e-6, (pen.energy, pen_ref.energy)
    for kern in ("items", "default"):
        a, b = out[kern, "1"], out[kern, "0"]
        assert a._sharded_stats["converged"] and b._sharded_stats["converged"], (kern, a._sharded_stats, b._sharded_stats)
        assert a._sharded_stats["n_allgather"] >= a._sharded_stats["n_sigma"], (kern, a._sharded_stats)
        assert a.energy == b.energy and np.array_equal(a.sci_state.amplitudes, b.sci_state.amplitudes), (kern, a.energy, b.energy)
        assert abs(a.energy - ref.energy) < 1e-8, (kern, a.energy, ref.energy)
        assert abs(abs(np.vdot(a.sci_state.amplitudes, ref.sci_state.amplitudes)) - 1.0) < 1e-8


def test_config2_full_size_1e4_x_1e4(hip_lib):
    """BASELINE config 2 read literally: N2-sized (16e,30o), 10^4 uniform-random strings per spin, D = 10^8 (800 MB
    per vector) on one MI355X.  Kernel selection; sigma rows against the row-restricted string-space oracle (O1s,
    `sigma_rows_string_space`) on a sampled subset of alpha rows -- first and last row, rows with alpha single links,
    random rows; hermiticity on two vectors; sigma bitwise reproducible; then ONE whole Davidson solve: converged,
    Ritz value = Rayleigh quotient of the returned state, true residual |Hc - Ec| re-evaluated with a separate sigma
    call, variational bounds (E0 <= min hdiag; E0 <= the energy of the start vector), observables consistent."""
    from qiskit_addon_sqd_amd import synthetic as S

    n = 10000
    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, n, 51), S.uniform_strings(30, 8, n, 52)
    rng = np.random.default_rng(10)
    x = rng.standard_normal((n, n), dtype=np.float32).astype(np.float64)
    y = rng.standard_normal((n, n), dtype=np.float32).astype(np.float64)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() in ("k_sigma_rows<2>", "k_sigma_lists")
        assert ctx.sigma_bytes() > 16.0 * n * n
        sx = ctx.sigma(x)
        sy = ctx.sigma(y)
        assert np.array_equal(sx, ctx.sigma(x))  # fixed summation order: the same bits on every run
        # --- sampled rows against the oracle
        sl = O.single_links(sa, 30)
        with_singles = np.unique(sl["tgt"])
        rows = np.unique(np.concatenate(([0, 1, n - 2, n - 1], with_singles[:6], with_singles[-6:],
                                         rng.choice(n, 12, replace=False))))
        ref = O.sigma_rows_string_space(h1, eri, sa, sb, x, 30, rows)
        hd = ctx.hdiag()
        hd_max = np.abs(hd).max()
        assert np.abs(sx[rows] - ref).max() < 1e-11 * hd_max * max(1.0, np.abs(x).max())
        # the same columns-wise: sigma of the TRANSPOSED problem (alpha <-> beta) on sampled rows = sampled columns here
        cols = np.unique(np.concatenate(([0, n - 1], rng.choice(n, 6, replace=False))))
        refT = O.sigma_rows_string_space(h1, eri, sb, sa, np.ascontiguousarray(x.T), 30, cols)
        assert np.abs(sx[:, cols].T - refT).max() < 1e-11 * hd_max * max(1.0, np.abs(x).max())
        # --- hermiticity
        a, b = np.vdot(y, sx), np.vdot(sy, x)
        assert abs(a - b) < 1e-9 * abs(a)
        del sy, y
        # --- one whole Davidson solve
        amps, st = ctx.davidson()
        assert st["converged"] == 1 and st["n_sigma"] >= 2
        e0 = st["e_davidson"]
        assert abs(np.vdot(amps, amps) - 1.0) < 1e-12
        hc = ctx.sigma(amps)
        assert abs(np.vdot(amps, hc) - e0) < 1e-9  # Ritz value = Rayleigh quotient of the returned vector
        resid = np.linalg.norm((hc - e0 * amps).ravel())
        assert resid < np.sqrt(1e-9) * 1.01 and abs(resid - st["residual"]) < 1e-7  # default rule: pyscf's |r| < sqrt(tol)
        assert e0 <= hd.min() + 1e-12  # variational: below the lowest diagonal element ...
        x0 = ctx.init_guess()
        assert e0 <= np.vdot(x0, ctx.sigma(x0)) + 1e-12  # ... and below the start vector's energy
        e, s2, oa, ob = ctx.observables()
        assert abs(e - e0) < 1e-9
        assert abs(oa.sum() - 8.0) < 1e-9 and abs(ob.sum() - 8.0) < 1e-9
        assert -1e-9 <= s2 <= 8.0 * 9.0 + 1e-9


@pytest.mark.parametrize("na,nb", [(1500, 1111), (1111, 2050), (700, 4097)])
def test_list_passes_forced_against_work_items(hip_lib, monkeypatch, na, nb):
    """The list passes (sqd_lists.hip; the default from 5 000 strings per spin) forced at sizes where the work-item kernel
    can be forced on the same inputs: ragged shapes (last column block and last row chunk partial, odd row lengths: the
    16-byte aligned image of a row starts one double early on every other row, unaligned tile stores), nalpha != nbeta,
    every operator form -- H, S^2 alone (kernel variant 3), the linear penalty (variant 2), the squared penalty -- and a
    Davidson solve with the spin penalty."""
    from qiskit_addon_sqd_amd import synthetic as S

    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, na, 31), S.uniform_strings(30, 7, nb, 32)
    x = np.random.default_rng(8).standard_normal((na, nb))
    out = {}
    for forced in ("lists", "items"):
        if forced == "lists":
            monkeypatch.setenv("SQD_SIGMA_LISTS", "1")
            monkeypatch.delenv("SQD_SIGMA_ROWS", raising=False)
            monkeypatch.delenv("SQD_SIGMA_DIRECT", raising=False)
        else:
            monkeypatch.setenv("SQD_SIGMA_LISTS", "0")
            monkeypatch.setenv("SQD_SIGMA_ROWS", "0")
            monkeypatch.setenv("SQD_SIGMA_DIRECT", "0")
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == ("k_sigma_lists" if forced == "lists" else "k_sigma")
            ops = (ctx.sigma(x), ctx.contract_ss(x), ctx.sigma(x, 1, 0.75, 0.3), ctx.sigma(x, 2, 0.75, 0.3))
            assert np.array_equal(ops[0], ctx.sigma(x))  # fixed summation order
            _, st = ctx.davidson(spin_sq=0.75, shift=0.3)
            out[forced] = ops + (st["e_davidson"], st["converged"], ctx.observables())
    a, b = out["lists"], out["items"]
    for u, v in zip(a[:4], b[:4]):
        assert np.abs(u - v).max() < 1e-12 * max(1.0, np.abs(v).max())
    assert a[5] == 1 and b[5] == 1 and abs(a[4] - b[4]) < 1e-9
    assert abs(a[6][0] - b[6][0]) < 1e-9 and abs(a[6][1] - b[6][1]) < 1e-8
    assert np.abs(a[6][2] - b[6][2]).max() < 1e-8 and np.abs(a[6][3] - b[6][3]).max() < 1e-8


def test_list_path_default_selection_mid_size(hip_lib, monkeypatch):
    """The list path where it becomes the DEFAULT (from 5 000 strings per spin; below, k_sigma_rows): 5 003 x 6 101
    uniform strings, nalpha != nbeta, odd row length (k_alpha_rows' 8-byte form, unaligned row images).  H, the
    linear spin penalty and S^2 alone against k_sigma_rows on the same input (SQD_SIGMA_LISTS=0); H on sampled rows and
    columns against the row-restricted string-space oracle."""
    from qiskit_addon_sqd_amd import synthetic as S

    na, nb = 5003, 6101
    h1, eri = S.synthetic_integrals(30)
    sa, sb = S.uniform_strings(30, 8, na, 61), S.uniform_strings(30, 8, nb, 62)
    rng = np.random.default_rng(12)
    x = rng.standard_normal((na, nb))
    out = {}
    for forced in ("default", "rows"):
        if forced == "rows":
            monkeypatch.setenv("SQD_SIGMA_LISTS", "0")
        else:
            monkeypatch.delenv("SQD_SIGMA_LISTS", raising=False)
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert (ctx.sigma_kernel() == "k_sigma_lists") == (forced == "default"), ctx.sigma_kernel()
            out[forced] = (ctx.sigma(x), ctx.sigma(x, 1, 0.75, 0.3), ctx.contract_ss(x))
            if forced == "default":
                assert np.array_equal(out[forced][0], ctx.sigma(x))  # fixed summation order
                hd_max = np.abs(ctx.hdiag()).max()
    for u, v in zip(out["default"], out["rows"]):
        assert np.abs(u - v).max() < 1e-12 * max(1.0, np.abs(v).max())
    sx = out["default"][0]
    sl = O.single_links(sa, 30)
    with_singles = np.unique(sl["tgt"])
    rows = np.unique(np.concatenate(([0, na - 1], with_singles[:4], with_singles[-4:], rng.choice(na, 8, replace=False))))
    ref = O.sigma_rows_string_space(h1, eri, sa, sb, x, 30, rows)
    assert np.abs(sx[rows] - ref).max() < 1e-11 * hd_max * max(1.0, np.abs(x).max())
    cols = np.unique(np.concatenate(([0, nb - 1], rng.choice(nb, 6, replace=False))))
    refT = O.sigma_rows_string_space(h1, eri, sb, sa, np.ascontiguousarray(x.T), 30, cols)
    assert np.abs(sx[:, cols].T - refT).max() < 1e-11 * hd_max * max(1.0, np.abs(x).max())


# ---- CONNECTED subspaces at D = 1e6 .. 1e7: what the SQD loop's carry-over produces and what
# the reference advertises ("subspace dimensions of ~1e7", /root/reference/README.md:78).  HF-centred sets thin out as
# they grow (same-spin blocks 11 % dense at 1000 strings, 5.6 % at 3000), and every same-spin formulation is run on the
# same inputs: the matrix cores (k_same_spin_mfma), the sparse work items, the sparse product on C and C^T (sqd_spmm.hip).
import functools  # noqa: E402

_CONNECTED_MODES = {
    "mfma": ({"SQD_SIGMA_DENSE": "1"}, "k_same_spin_mfma+k_sigma"),
    "sparse": ({"SQD_SIGMA_DENSE": "0"}, "k_sigma"),
    "spmm": ({"SQD_SIGMA_SPMM": "1"}, "k_spmm_grouped+k_opp_rows"),
    "spmm_items": ({"SQD_SIGMA_SPMM": "1", "SQD_SIGMA_OPP": "0"}, "k_spmm_grouped+k_sigma"),
}


def _set_mode(monkeypatch, mode):
    for k in ("SQD_SIGMA_DENSE", "SQD_SIGMA_SPMM", "SQD_SIGMA_OPP"):
        monkeypatch.delenv(k, raising=False)
    env, kernel = _CONNECTED_MODES[mode]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    return kernel


@functools.lru_cache(maxsize=None)
def _connected_reference_1000():
    """Oracle side of the 1000 x 1000 test, computed once for all modes: the full string-space sigma of a random vector
    and the oracle's own Davidson (pyscf control flow) on the string-space operator (~3 minutes of numpy)."""
    n, norb = 1000, 30
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, n, 11), O.hf_centred_strings(norb, 8, n, 13)
    x = np.random.default_rng(17).standard_normal((n, n))
    ref = O.sigma_string_space(h1, eri, sa, sb, x, norb)
    hd = O.make_hdiag(h1, eri, sa, sb, norb)
    op = O.StringSpaceOperator(h1, eri, sa, sb, norb)
    conv, e_ref, x_ref, _ = O.davidson_pyscf(op, O.init_guess(hd.ravel(), n, n, nelec=(8, 8)), hd.ravel(), tol=1e-12, max_cycle=200)  # (|r| < 1e-6: the occupancies below are first order in it)
    assert conv
    return h1, eri, sa, sb, x, ref, hd, float(e_ref), x_ref / np.linalg.norm(x_ref)


@pytest.mark.parametrize("mode", ["mfma", "spmm", "spmm_items", "sparse"])
def test_connected_1000x1000_full_sigma_and_solve(hip_lib, monkeypatch, mode):
    """HF-centred 1000 x 1000 (D = 1e6, ~10 single + ~100 double links per string): the FULL sigma vector and hdiag
    against O1s, bitwise reproducibility, then one whole solve -- E0 against the oracle's own Davidson (1e-8 Ha; the
    north_star bar is 1e-6), the state's overlap, occupancies -- with the default residual rule (pyscf's) for the
    energy and the tight rule for the first-order quantities."""
    kernel = _set_mode(monkeypatch, mode)
    h1, eri, sa, sb, x, ref, hd_ref, e_ref, x_ref = _connected_reference_1000()
    n = len(sa)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == kernel
        hd = ctx.hdiag()
        scale = np.abs(hd).max()
        assert np.allclose(hd, hd_ref, rtol=0, atol=1e-11 * scale)
        sx = ctx.sigma(x)
        assert np.abs(sx - ref).max() < 1e-11 * scale * max(1.0, np.abs(x).max())
        assert np.array_equal(sx, ctx.sigma(x))  # fixed summation order: the same bits on every run
        _, st = ctx.davidson(fetch=False)  # the default rule: |r| < sqrt(tol) without a penalty (pyscf's)
        assert st["converged"] == 1 and abs(st["e_davidson"] - e_ref) < 1e-8 and st["residual"] < np.sqrt(1e-9)
        amps, st2 = ctx.davidson(tol_residual=np.sqrt(1e-9) / 32.0)
        assert st2["converged"] == 1 and abs(st2["e_davidson"] - e_ref) < 1e-8 and st2["n_sigma"] >= st["n_sigma"]
        assert abs(abs(np.vdot(amps.ravel(), x_ref)) - 1.0) < 1e-9
        e, s2, occ_a, occ_b = ctx.observables()
        assert abs(e - e_ref) < 1e-8
    r1a, r1b = O.make_rdm1s(x_ref.reshape(n, n), sa, sb, 30)
    assert np.allclose(occ_a, np.diag(r1a), atol=5e-6) and np.allclose(occ_b, np.diag(r1b), atol=5e-6)


@functools.lru_cache(maxsize=None)
def _connected_reference_3000():
    """Oracle side of the 3000 x 3000 test: sampled rows and sampled columns of the string-space sigma of a random vector."""
    n, norb = 3000, 30
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, n, 11), O.hf_centred_strings(norb, 8, n, 13)
    rng = np.random.default_rng(19)
    x = rng.standard_normal((n, n), dtype=np.float32).astype(np.float64)
    # rows / columns: the Hartree-Fock string itself (the longest lists: 176 single links), its neighbourhood, the last
    # strings (highest excitation rank: the shortest lists), random ones
    hf = int(np.flatnonzero(sa == (1 << 8) - 1)[0])
    rows = np.unique(np.concatenate(([0, 1, hf, n - 2, n - 1], rng.choice(n, 9, replace=False))))
    hfb = int(np.flatnonzero(sb == (1 << 8) - 1)[0])
    cols = np.unique(np.concatenate(([0, hfb, n - 1], rng.choice(n, 7, replace=False))))
    ref_rows = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
    ref_cols = O.sigma_rows_string_space(h1, eri, sb, sa, np.ascontiguousarray(x.T), norb, cols)
    return h1, eri, sa, sb, x, rows, cols, ref_rows, ref_cols


@pytest.mark.parametrize("mode", ["spmm", "spmm_items", "mfma", "sparse"])
def test_connected_3000x3000_sampled_sigma_and_solve(hip_lib, monkeypatch, mode):
    """HF-centred 3000 x 3000 (D = 9e6, 72 MB per vector; ~11 single + ~155 double links per string, same-spin blocks
    5.6 % dense -- the default here is the sparse product of sqd_spmm.hip): sigma on sampled rows AND sampled columns
    against the row-restricted string-space oracle, hermiticity, bitwise reproducibility; then one whole solve: converged,
    Ritz value = Rayleigh quotient of the returned state (separate sigma call), true residual below the stopping rule,
    variational bounds, observables consistent.  Every same-spin formulation gives the same energy."""
    kernel = _set_mode(monkeypatch, mode)
    h1, eri, sa, sb, x, rows, cols, ref_rows, ref_cols = _connected_reference_3000()
    n = len(sa)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        if mode == "spmm":  # (what the default selection takes at this size)
            monkeypatch.delenv("SQD_SIGMA_SPMM")
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == kernel
        hd = ctx.hdiag()
        scale = np.abs(hd).max() * max(1.0, np.abs(x).max())
        sx = ctx.sigma(x)
        assert np.abs(sx[rows] - ref_rows).max() < 1e-11 * scale
        assert np.abs(sx[:, cols].T - ref_cols).max() < 1e-11 * scale
        assert np.array_equal(sx, ctx.sigma(x))
        y = np.random.default_rng(23).standard_normal((n, n), dtype=np.float32).astype(np.float64)
        a, b = np.vdot(y, sx), np.vdot(ctx.sigma(y), x)
        assert abs(a - b) < 1e-9 * abs(a)
        del y, sx
        amps, st = ctx.davidson()
        assert st["converged"] == 1 and st["n_sigma"] >= 10
        e0 = st["e_davidson"]
        assert abs(np.vdot(amps, amps) - 1.0) < 1e-12
        hc = ctx.sigma(amps)
        assert abs(np.vdot(amps, hc) - e0) < 1e-9
        resid = np.linalg.norm((hc - e0 * amps).ravel())
        assert resid < np.sqrt(1e-9) * 1.01 and abs(resid - st["residual"]) < 1e-7  # default rule: pyscf's sqrt(tol)
        assert e0 <= hd.min() + 1e-12
        x0 = ctx.init_guess()
        assert e0 <= np.vdot(x0, ctx.sigma(x0)) + 1e-12
        e, s2, oa, ob = ctx.observables()
        assert abs(e - e0) < 1e-9 and abs(oa.sum() - 8.0) < 1e-9 and abs(ob.sum() - 8.0) < 1e-9
    _CONNECTED_E0.setdefault(3000, e0)
    assert abs(e0 - _CONNECTED_E0[3000]) < 1e-8  # the same ground state whichever way the same-spin part is formed


_CONNECTED_E0: dict = {}


def test_rdm2_row_form_on_a_connected_set(hip_lib, monkeypatch):
    """rdm2 of a state on HF-centred 1000 x 1100 (3e9 opposite-spin link pairs: the row form of sqd_rdm.hip by default)
    against the thread-per-link form of rounds 1-4 on the same state, the resolved blocks, the trace, and E = <c|H|c>
    from the RDMs contracted as the reference does (fermion.py:730-732)."""
    norb = 30
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, 1000, 11), O.hf_centred_strings(norb, 7, 1100, 13)
    amps = np.random.default_rng(3).standard_normal((len(sa), len(sb))) * np.exp(-np.add.outer(np.arange(len(sa)), np.arange(len(sb))) / 400.0)
    amps /= np.linalg.norm(amps)
    out = {}
    for hook in (None, "0"):
        if hook is None:
            monkeypatch.delenv("SQD_RDM2_ROWS", raising=False)
        else:
            monkeypatch.setenv("SQD_RDM2_ROWS", hook)
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            out[hook] = ctx.rdm2(amps)
            if hook is None:
                aa, ab, bb = ctx.rdm2s(amps)
                d1a, d1b = ctx.rdm1s(amps)
                e_sigma = float(np.vdot(amps, ctx.sigma(amps)))
    d2 = out[None]
    assert np.abs(d2 - out["0"]).max() < 1e-12
    assert np.abs(aa + bb + ab + ab.transpose(2, 3, 0, 1) - d2).max() < 1e-12
    assert abs(np.einsum("ppqq->", d2) - 15.0 * 14.0) < 1e-9
    e_rdm = np.einsum("pq,pq->", h1, d1a + d1b) + 0.5 * np.einsum("prqs,prqs->", eri, d2)
    assert abs(e_rdm - e_sigma) < 1e-9 * max(1.0, abs(e_sigma))



@pytest.mark.timeout(300)
@pytest.mark.parametrize("na,nb", [(900, 8193), (1000, 10000)])
def test_connected_more_than_8192_beta_strings(hip_lib, monkeypatch, na, nb):
    """HF-centred sets with more than 8192 beta strings (nine and more columns per thread in the work-item kernel, which
    runs behind the sparse product there: the whole-row kernels do not take such rows).  Round 5's single-pass
    instantiation k_sigma<16, ., true, false> did not return on the MI355X at nb = 8193: its body was an out-of-line
    function whose long branches overwrote the return address (profiles/r06/hang_root_cause.txt; fixed by inlining,
    pinned by tests/test_codeobj_audit.py and test_single_pass_r16_instantiation_returns below).  Sigma on sampled rows
    against the row-restricted string-space oracle, the work items alone on the whole vector, and it comes back."""
    for k in ("SQD_SIGMA_DENSE", "SQD_SIGMA_SPMM", "SQD_SIGMA_OPP"):
        monkeypatch.delenv(k, raising=False)
    norb = 30
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, na, 31), O.hf_centred_strings(norb, 8, nb, 37)
    rng = np.random.default_rng(43)
    x = rng.standard_normal((na, nb), dtype=np.float32).astype(np.float64)
    rows = np.unique(np.concatenate(([0, na - 1], rng.choice(na, 3, replace=False))))
    ref_rows = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_spmm_grouped+k_sigma"
        scale = np.abs(ctx.hdiag()).max() * max(1.0, np.abs(x).max())
        sx = ctx.sigma(x)
        assert np.abs(sx[rows] - ref_rows).max() < 1e-11 * scale
        assert np.array_equal(sx, ctx.sigma(x))
        assert np.abs(ctx.sigma(x, use_spin=1, ss=0.75, shift=0.3)[rows]).max() > 0.0  # (the penalty form returns as well)
    monkeypatch.setenv("SQD_SIGMA_SPMM", "0")
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_sigma"
        assert np.abs(ctx.sigma(x) - sx).max() < 1e-11 * scale



def test_single_pass_r16_instantiation_returns(hip_lib, tmp_path):
    """The launch that never returned in round 5, k_sigma<16, false, true, false> (rows of nine and more columns per thread
    whose beta lists fit one pass), forced back on in a child process (SQD_SIGMA_R16_SINGLE=1 is read once per process)
    under a timeout: it returns, and with the bits of the default routing (the multi-pass instantiation with zero extra
    passes evaluates the same sums in the same order)."""
    import subprocess
    import sys as _sys

    script = tmp_path / "r16.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "from oracle import sqd_oracle as O\n"
        "from qiskit_addon_sqd_amd import _capi\n"
        "h1, eri = O.synthetic_integrals(30)\n"
        "sa, sb = O.hf_centred_strings(30, 8, 300, 31), O.hf_centred_strings(30, 8, 8193, 37)\n"
        "x = np.random.default_rng(43).standard_normal((300, 8193), dtype=np.float32).astype(np.float64)\n"
        "with _capi.Context(h1, eri) as ctx:\n"
        "    ctx.set_subspace(sa, sb)\n"
        "    np.save(sys.argv[1], ctx.sigma(x))\n"
    )
    outs = []
    for tag, hook in (("multi", "0"), ("single", "1")):
        env = dict(os.environ, SQD_SIGMA_R16_SINGLE=hook, SQD_SIGMA_OPP="0", SQD_SIGMA_SPMM="0", SQD_SIGMA_DENSE="0")
        out = tmp_path / f"{tag}.npy"
        subprocess.run([_sys.executable, str(script), str(out)], env=env, check=True, timeout=180)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0]).max() > 0.0


@pytest.mark.parametrize("na,nb,src", [(1000, 5003, None), (901, 3500, None), (700, 7300, None), (901, 2500, "1")])
def test_connected_long_rows_source_range_kernel(hip_lib, monkeypatch, na, nb, src):
    """Rows of more than 3072 columns: k_opp_src (sqd_oppsrc.hip: passes over ranges of the source column, round 6; the
    4-8-columns-per-thread instantiations of k_opp_rows that used to run them spilled 6-40 registers).  HF-centred
    1000 x 5003, 901 x 3500 and 700 x 7300 (more than six columns per thread: one sub-run) by default selection, 901 x 2500
    forced (SQD_OPP_SRC=1: 512 threads, the 32 KB layout) -- sigma on sampled rows and sampled columns against the
    row-restricted string-space oracle, reproducibility, the linear spin penalty and the plain operator against the
    work-item formulation on the whole vector, one whole solve (Rayleigh quotient, residual)."""
    for k in ("SQD_SIGMA_DENSE", "SQD_SIGMA_SPMM", "SQD_SIGMA_OPP", "SQD_OPP_SRC"):
        monkeypatch.delenv(k, raising=False)
    if src is not None:
        monkeypatch.setenv("SQD_OPP_SRC", src)
    norb = 30
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, 8, na, 31), O.hf_centred_strings(norb, 8, nb, 37)
    rng = np.random.default_rng(41)
    x = rng.standard_normal((na, nb), dtype=np.float32).astype(np.float64)
    hf = int(np.flatnonzero(sa == (1 << 8) - 1)[0])
    rows = np.unique(np.concatenate(([0, hf, na - 1], rng.choice(na, 4, replace=False))))
    hfb = int(np.flatnonzero(sb == (1 << 8) - 1)[0])
    cols = np.unique(np.concatenate(([0, hfb, nb - 1, min(3071, nb - 1), min(3072, nb - 1), min(3073, nb - 1)], rng.choice(nb, 3, replace=False))))
    ref_rows = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
    ref_cols = O.sigma_rows_string_space(h1, eri, sb, sa, np.ascontiguousarray(x.T), norb, cols)
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_spmm_grouped+k_opp_src"
        scale = np.abs(ctx.hdiag()).max() * max(1.0, np.abs(x).max())
        sx = ctx.sigma(x)
        assert np.abs(sx[rows] - ref_rows).max() < 1e-11 * scale
        assert np.abs(sx[:, cols].T - ref_cols).max() < 1e-11 * scale
        assert np.array_equal(sx, ctx.sigma(x))
        sp = ctx.sigma(x, use_spin=1, ss=0.75, shift=0.3)  # (the linear penalty form: the same kernel, one weight per entry shifted)
        amps, st = ctx.davidson()
        e0 = st["e_davidson"]
        hc = ctx.sigma(amps)
        assert st["converged"] == 1 and abs(np.vdot(amps, hc) - e0) < 1e-9
        assert np.linalg.norm((hc - e0 * amps).ravel()) < np.sqrt(1e-9) * 1.01
    monkeypatch.setenv("SQD_SIGMA_SPMM", "0")
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
    with _capi.Context(h1, eri, lib=hip_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_sigma"
        assert np.abs(ctx.sigma(x) - sx).max() < 1e-11 * scale
        assert np.abs(ctx.sigma(x, use_spin=1, ss=0.75, shift=0.3) - sp).max() < 1e-11 * scale



def test_connected_ragged_fes_sized_sparse_product_path(hip_lib, monkeypatch):
    """The sparse-product + whole-row path away from the round numbers it was tuned on: Fe-S-sized orbital space (40
    orbitals: 820 orbital pairs per weight row), nalpha != nbeta electrons (15, 14) and strings (1153 x 931: ragged row
    groups, panels, column ranges and transposition tiles), HF-centred.  Kernel selection by default; H on sampled rows and
    columns against the row-restricted string-space oracle; H + the linear spin penalty and S^2 (work items behind the
    product) against the forced work-item kernel; one whole solve against the work-item solve."""
    norb, nelec = 40, (15, 14)
    h1, eri = O.synthetic_integrals(norb)
    sa, sb = O.hf_centred_strings(norb, nelec[0], 1153, 21), O.hf_centred_strings(norb, nelec[1], 931, 22)
    na, nb = len(sa), len(sb)
    rng = np.random.default_rng(29)
    x = rng.standard_normal((na, nb))
    out = {}
    for forced in ("default", "items"):
        for k in ("SQD_SIGMA_DENSE", "SQD_SIGMA_SPMM", "SQD_SIGMA_OPP"):
            monkeypatch.delenv(k, raising=False)
        if forced == "items":
            monkeypatch.setenv("SQD_SIGMA_DENSE", "0")
            monkeypatch.setenv("SQD_SIGMA_SPMM", "0")
        with _capi.Context(h1, eri, lib=hip_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == ("k_spmm_grouped+k_opp_rows" if forced == "default" else "k_sigma"), ctx.sigma_kernel()
            ops = (ctx.sigma(x), ctx.sigma(x, 1, 0.75, 0.3), ctx.contract_ss(x))
            assert np.array_equal(ops[0], ctx.sigma(x))
            _, st = ctx.davidson(fetch=False)
            out[forced] = ops + (st["e_davidson"], st["converged"], np.abs(ctx.hdiag()).max())
    a, b = out["default"], out["items"]
    scale = a[5] * max(1.0, np.abs(x).max())
    for u, v in zip(a[:3], b[:3]):
        assert np.abs(u - v).max() < 1e-11 * scale
    assert a[4] == 1 and b[4] == 1 and abs(a[3] - b[3]) < 1e-8
    hfa = int(np.flatnonzero(sa == (1 << nelec[0]) - 1)[0])
    rows = np.unique(np.concatenate(([0, hfa, na - 1], rng.choice(na, 7, replace=False))))
    ref = O.sigma_rows_string_space(h1, eri, sa, sb, x, norb, rows)
    assert np.abs(a[0][rows] - ref).max() < 1e-11 * scale
    cols = np.unique(np.concatenate(([0, nb - 1], rng.choice(nb, 6, replace=False))))
    refT = O.sigma_rows_string_space(h1, eri, sb, sa, np.ascontiguousarray(x.T), norb, cols)
    assert np.abs(a[0][:, cols].T - refT).max() < 1e-11 * scale
