"""Analytic, pyscf-free anchors of the floating-point layer (the reference's own
tests build their expectations with pyscf at run time, and pyscf is in neither container).  Closed forms that pin the
conventions the HIP kernels share with pyscf -- chemist-order ``eri`` and its 1/2 factor, the ``rdm2`` index order that the
reference contracts with ``"prqs,prqs"`` (``/root/reference/qiskit_addon_sqd/fermion.py:730-732``), occupancies by orbital =
bit position -- through the product's native path: the emulator build in the CPU suite, the hipcc build on the GPU."""
import numpy as np
import pytest

from qiskit_addon_sqd_amd import _capi


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def lib(request):
    return request.getfixturevalue("emu_lib" if request.param == "emu" else "hip_lib")


def _all_strings(norb, nocc):
    return np.array(sorted(s for s in range(1 << norb) if bin(s).count("1") == nocc), dtype=np.int64)


@pytest.mark.parametrize("t,U", [(1.0, 4.0), (0.7, 0.0), (0.25, 11.5)])
def test_two_site_hubbard_closed_form(lib, t, U):
    """Half-filled Hubbard dimer: E0 = U/2 - sqrt(U^2/4 + 4 t^2).  h1 = -t (hopping), chemist-order eri[p,p,p,p] = U:
    an ``eri`` read in physicist order, or a missing 1/2 in front of the two-body term, moves E0 at O(U)."""
    h1 = np.array([[0.0, -t], [-t, 0.0]])
    eri = np.zeros((2, 2, 2, 2))
    eri[0, 0, 0, 0] = eri[1, 1, 1, 1] = U
    s = _all_strings(2, 1)
    with _capi.Context(h1, eri, lib=lib) as ctx:
        ctx.set_subspace(s, s)
        amps, st = ctx.davidson(tol=1e-12, tol_residual=1e-7)
        e, s2, oa, ob = ctx.observables()
        d1a, d1b = ctx.rdm1s()
        d2 = ctx.rdm2()
    e_exact = 0.5 * U - np.sqrt(0.25 * U * U + 4.0 * t * t)
    assert abs(st["e_davidson"] - e_exact) < 1e-12 and abs(e - e_exact) < 1e-12
    assert abs(s2) < 1e-10  # the ground state is the singlet
    assert np.allclose(oa, 0.5, atol=1e-10) and np.allclose(ob, 0.5, atol=1e-10)  # site symmetry
    # double occupancy <n_up n_dn> of a site in closed form, read off rdm2 in pyscf's index order dm2[p,p,p,p]
    docc = 0.25 * (1.0 - (U / 4.0) / np.sqrt((U / 4.0) ** 2 + t * t)) if (U or t) else 0.25
    assert abs(0.5 * d2[0, 0, 0, 0] - docc) < 1e-10 and abs(0.5 * d2[1, 1, 1, 1] - docc) < 1e-10
    # the reference's energy expression on these RDMs (fermion.py:730-732)
    e_rdm = np.einsum("pr,pr->", d1a + d1b, h1) + 0.5 * np.einsum("prqs,prqs->", d2, eri)
    assert abs(e_rdm - e_exact) < 1e-11


def test_zero_eri_fci_limit(lib):
    """eri = 0 in the complete string space: E0 = the sum of the lowest orbital energies of each spin, the 1-RDMs are
    idempotent projectors onto those orbitals, rdm2 factorises."""
    norb, na, nb = 5, 2, 3
    rng = np.random.default_rng(41)
    h1 = rng.standard_normal((norb, norb))
    h1 = np.diag(np.arange(norb) - 2.0) + 0.2 * (h1 + h1.T)  # (orbital energies ~1 apart: the diagonal preconditioner works)
    eri = np.zeros((norb,) * 4)
    sa, sb = _all_strings(norb, na), _all_strings(norb, nb)
    with _capi.Context(h1, eri, lib=lib) as ctx:
        ctx.set_subspace(sa, sb)
        amps, st = ctx.davidson(tol=1e-10, tol_residual=1e-7)  # (|t|^2 < lindep = 1e-14 ends a run, as in pyscf: no tighter)
        e, s2, oa, ob = ctx.observables()
        d1a, d1b = ctx.rdm1s()
        d2 = ctx.rdm2()
    w, v = np.linalg.eigh(h1)
    assert st["converged"] == 1
    assert abs(e - (w[:na].sum() + w[:nb].sum())) < 1e-11
    pa, pb = v[:, :na] @ v[:, :na].T, v[:, :nb] @ v[:, :nb].T
    assert np.allclose(d1a, pa, atol=1e-6) and np.allclose(d1b, pb, atol=1e-6)
    assert np.allclose(d1a @ d1a, d1a, atol=1e-6) and np.allclose(d1b @ d1b, d1b, atol=1e-6)
    assert np.allclose(oa, np.diag(pa), atol=1e-6) and np.allclose(ob, np.diag(pb), atol=1e-6)
    # a single determinant in the rotated basis: dm2[pqrs] = g_pq g_rs - sum_sigma g^s_ps g^s_rq (pyscf's index order)
    g = pa + pb
    ref = np.einsum("pq,rs->pqrs", g, g) - np.einsum("ps,rq->pqrs", pa, pa) - np.einsum("ps,rq->pqrs", pb, pb)
    assert np.allclose(d2, ref, atol=1e-5)


def test_rdm2_index_order_single_determinant(lib):
    """rdm2 of ONE determinant against the textbook form
        dm2[p,q,r,s] = sum_{sigma tau} <p+_sigma r+_tau s_tau q_sigma> = g_pq g_rs - sum_sigma g^sigma_ps g^sigma_rq,
    element by element, and contracted the reference's way -- einsum("prqs,prqs") as written in fermion.py:730-732 --
    with a probe tensor that has NONE of the 8-fold symmetries of a real ``eri``: an index transposition in the kernels
    (p <-> q, (pq) <-> (rs), ...) cannot hide behind the symmetry of the integrals."""
    norb = 6
    sa, sb = np.array([0b010110], dtype=np.int64), np.array([0b101001], dtype=np.int64)  # alpha {1,2,4}, beta {0,3,5}
    rng = np.random.default_rng(43)
    h1 = rng.standard_normal((norb, norb))
    h1 = 0.5 * (h1 + h1.T)
    eri = np.zeros((norb,) * 4)
    with _capi.Context(h1, eri, lib=lib) as ctx:
        ctx.set_subspace(sa, sb)
        amps = np.ones((1, 1))
        d1a, d1b = ctx.rdm1s(amps)
        d2 = ctx.rdm2(amps)
        d2aa, d2ab, d2bb = ctx.rdm2s(amps)
    ga = np.diag([float((int(sa[0]) >> p) & 1) for p in range(norb)])
    gb = np.diag([float((int(sb[0]) >> p) & 1) for p in range(norb)])
    assert np.array_equal(d1a, ga) and np.array_equal(d1b, gb)  # occupancy of orbital p = bit p (LSB = orbital 0)
    g = ga + gb
    ref = np.einsum("pq,rs->pqrs", g, g) - np.einsum("ps,rq->pqrs", ga, ga) - np.einsum("ps,rq->pqrs", gb, gb)
    assert np.abs(d2 - ref).max() < 1e-12
    # spin blocks (pyscf make_rdm2s): aa / bb antisymmetrised, ab a plain product
    assert np.abs(d2aa - (np.einsum("pq,rs->pqrs", ga, ga) - np.einsum("ps,rq->pqrs", ga, ga))).max() < 1e-12
    assert np.abs(d2bb - (np.einsum("pq,rs->pqrs", gb, gb) - np.einsum("ps,rq->pqrs", gb, gb))).max() < 1e-12
    assert np.abs(d2ab - np.einsum("pq,rs->pqrs", ga, gb)).max() < 1e-12
    probe = rng.standard_normal((norb,) * 4)  # no symmetry at all
    lhs = np.einsum("prqs,prqs->", d2, probe)
    rhs = np.einsum("prqs,prqs->", ref, probe)
    assert abs(lhs - rhs) < 1e-11
    # ... and the transposed readings of the same tensor give DIFFERENT numbers with this probe (the check has teeth)
    for perm in ("qprs", "pqsr", "psrq"):  # (not "rspq": that one is a true symmetry of any 2-RDM)
        assert abs(np.einsum(f"{perm},pqrs->", ref, probe) - np.einsum("pqrs,pqrs->", ref, probe)) > 1e-3
