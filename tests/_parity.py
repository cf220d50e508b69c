"""Shared parity checks: the native path (real HIP library on the GPU box, or the kernel-logic
emulator on CPU) against the numpy oracle on the same seeded inputs."""
import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi


def make_problem(norb, nelec, na, nb, seed, hf=False):
    h1, eri = O.synthetic_integrals(norb, seed=seed)
    gen = O.hf_centred_strings if hf else O.random_strings
    sa = gen(norb, nelec[0], na, seed + 1)
    sb = gen(norb, nelec[1], nb, seed + 2)
    return h1, eri, sa, sb


def check_link_tables(ctx, sa, sb, norb, h1=None, eri=None):
    """Bit-exact CI-string addressing: targets, sources, orbitals, pair indices, signs."""
    for spin, strs in enumerate((sa, sb)):
        sl = O.single_links(strs, norb)
        g = ctx.single_links(spin)
        assert len(g["tgt"]) == len(sl["tgt"])
        for ko, kg in (("tgt", "tgt"), ("src", "src"), ("p", "cre"), ("q", "des"), ("sign", "sign")):
            assert np.array_equal(sl[ko], g[kg]), (spin, ko)
        assert np.array_equal(O.pair_index(sl["p"], sl["q"]), g["pair"])
        dl = O.double_links(strs, norb)
        g2 = ctx.double_links(spin)
        assert len(g2["tgt"]) == len(dl["tgt"])
        for k in ("tgt", "src", "sign"):
            assert np.array_equal(dl[k], g2[k]), (spin, k)
        assert np.array_equal(np.stack([dl["p"], dl["r"], dl["q"], dl["s"]], 1).reshape(-1, 4), g2["orbs"])
        if eri is not None and len(dl["tgt"]):
            p, r, q, s = dl["p"], dl["r"], dl["q"], dl["s"]
            ref = dl["sign"] * (eri[p, q, r, s] - eri[p, s, r, q])
            assert np.allclose(g2["value"], ref, rtol=0, atol=1e-13)


def check_operators(ctx, h1, eri, sa, sb, norb, nelec, rng, tol=1e-11):
    H = O.build_php(h1, eri, sa, sb, norb)
    S2 = O.build_spin_square(sa, sb, norb, nelec)
    D = len(sa) * len(sb)
    scale = max(1.0, np.abs(H).max())
    assert np.allclose(ctx.hdiag().ravel(), np.diag(H), rtol=0, atol=tol * scale)
    c = rng.standard_normal((len(sa), len(sb)))
    assert np.allclose(ctx.sigma(c).ravel(), H @ c.ravel(), rtol=0, atol=tol * scale * np.sqrt(D))
    assert np.allclose(ctx.contract_ss(c).ravel(), S2 @ c.ravel(), rtol=0, atol=tol * np.sqrt(D) * 10)
    ss, shift = 0.75, 0.3
    ref = (H + shift * (S2 - ss * np.eye(D))) @ c.ravel()
    assert np.allclose(ctx.sigma(c, 1, ss, shift).ravel(), ref, rtol=0, atol=tol * scale * np.sqrt(D))
    P = S2 - 2.0 * np.eye(D)
    ref = (H + shift * (P @ P)) @ c.ravel()
    assert np.allclose(ctx.sigma(c, 2, 2.0, shift).ravel(), ref, rtol=0, atol=10 * tol * scale * np.sqrt(D))
    return H, S2


def check_ground_state(ctx, H, S2, h1, eri, sa, sb, norb, e_tol=1e-8, with_rdm2=True, variants=True):
    # the start vector is pyscf's get_init_guess (lower-triangle rule included), normalised
    x0 = O.init_guess(np.diag(H), len(sa), len(sb), nelec=ctx.nelec)
    assert np.allclose(ctx.init_guess().ravel(), x0 / np.linalg.norm(x0), rtol=0, atol=1e-15)
    amps, st = ctx.davidson()  # first run after set_subspace: state block + start vector prepared by the table build
    amps_b, st_b = ctx.davidson()  # second run: the solver's own k_init_guess launch -- the same bits either way
    assert np.array_equal(amps, amps_b) and st["n_sigma"] == st_b["n_sigma"] and st["e_davidson"] == st_b["e_davidson"]
    amps_c, st_c, obs_c = ctx.solve(sa, sb)  # the one-call path (state written to the page-locked buffer by a kernel)
    assert np.array_equal(amps, amps_c) and st_c["e_davidson"] == st["e_davidson"]
    amps_d, st_d, obs_d = ctx.solve(sa, sb, pageable_result=True)  # ... and through the copy stream into pageable memory
    assert np.array_equal(amps, amps_d) and obs_c[0] == obs_d[0] and obs_c[1] == obs_d[1]
    assert np.array_equal(obs_c[2], obs_d[2]) and np.array_equal(obs_c[3], obs_d[3])
    w, v = np.linalg.eigh(H)
    assert st["converged"] == 1
    assert abs(st["e_davidson"] - w[0]) < e_tol
    assert abs(ctx.energy() - w[0]) < e_tol
    gap = w[1] - w[0] if len(w) > 1 else 1.0
    if gap > 1e-4:
        assert abs(abs(amps.ravel() @ v[:, 0]) - 1.0) < 1e-5
    assert abs(ctx.spin_square() - amps.ravel() @ S2 @ amps.ravel()) < 1e-9
    d1a, d1b = ctx.rdm1s()
    r1a, r1b = O.make_rdm1s(amps, sa, sb, norb)
    assert np.allclose(d1a, r1a, atol=1e-12) and np.allclose(d1b, r1b, atol=1e-12)
    assert abs(np.trace(d1a) - bin(int(sa[0])).count("1")) < 1e-10
    e_o, s2_o, oa, ob = ctx.observables()  # fused path used by solve_fermion
    assert abs(e_o - ctx.energy()) < 1e-11 and abs(s2_o - ctx.spin_square()) < 1e-11
    assert np.allclose(oa, np.diag(d1a), atol=1e-12) and np.allclose(ob, np.diag(d1b), atol=1e-12)
    if with_rdm2:
        d2 = ctx.rdm2()
        assert np.allclose(d2, O.make_rdm2(amps, sa, sb, norb), atol=1e-12)
        e_rdm = O.energy_from_rdms(h1, eri, d1a + d1b, d2)
        assert abs(e_rdm - ctx.energy()) < 1e-10
        n = np.trace(d1a) + np.trace(d1b)
        assert abs(np.einsum("ppqq->", d2) - n * (n - 1)) < 1e-9
        aa, ab, bb = ctx.rdm2s()  # pyscf make_rdm2s pieces (SCIState.rdm(2, spin_summed=False))
        assert np.allclose(aa + bb + ab + ab.transpose(2, 3, 0, 1), d2, atol=1e-12)
        if norb <= 7:
            for got, ref in zip((aa, ab, bb), O.jw_rdm2s(amps, sa, sb, norb)):
                assert np.allclose(got, ref, atol=1e-12)
    if variants:
        check_solver_variants(ctx, w, v, amps, st, e_tol)
    return amps, st


def check_solver_variants(ctx, w, v, amps, st, e_tol):
    """The same eigenpair through the other control-flow paths of the device Davidson: fused native call,
    un-normalised user start vector, frequent restarts (max_space 2/3), a long basis (max_space 20, the
    wide-register kernels), the cycle limit, and bitwise run-to-run reproducibility."""
    D = amps.size
    # fused call = Davidson + observables
    a2, st2, (e2, s2_2, oa2, ob2) = ctx.davidson(observables=True)
    assert np.array_equal(a2, amps) and st2["n_sigma"] == st["n_sigma"]  # reproducible to the bit
    assert abs(e2 - w[0]) < e_tol and abs(oa2.sum() - round(oa2.sum())) < 1e-9
    # a user vector far from normalised (the norm is measured by the first fused reduction, never applied)
    rng = np.random.default_rng(D)
    ci0 = 37.5 * (amps + 0.05 * rng.standard_normal(amps.shape))
    # (cycle limit well above the default 100: the 70 x 70 FCI case needs ~90-100 iterations from either start,
    # and which side of 100 it lands on depends on the last bits of sigma)
    a3, st3 = ctx.davidson(ci0, max_cycle=400)
    assert st3["converged"] == 1 and abs(st3["e_davidson"] - w[0]) < e_tol
    assert abs(np.linalg.norm(a3) - 1.0) < 1e-9
    if D > 1:
        with pytest.raises(Exception, match="zero norm"):
            ctx.davidson(np.zeros_like(amps))
    for ms in (2, 3, 6, 20):
        _, stm = ctx.davidson(max_space=ms, max_cycle=400)
        if ms <= 3:
            # a restart every (other) iteration is close to preconditioned steepest descent: it may crawl on
            # hard cases (400 cycles are not enough for the 70 x 70 FCI problem), but it stays variational
            assert stm["e_davidson"] - w[0] > -1e-9 and (stm["converged"] == 0 or stm["e_davidson"] - w[0] < e_tol), stm
        else:
            assert stm["converged"] == 1 and abs(stm["e_davidson"] - w[0]) < e_tol, (ms, stm)
    if st["n_sigma"] > 2:
        a5, st5 = ctx.davidson(max_cycle=1)
        assert st5["converged"] == 0 and st5["n_sigma"] == 1 and st5["iterations"] == 1
        assert abs(np.linalg.norm(a5) - 1.0) < 1e-9
    # leave the context holding the reference solution again
    ctx.davidson()


def run_full_parity(lib, norb, nelec, na, nb, seed, hf=False, with_rdm2=True, variants=True):
    h1, eri, sa, sb = make_problem(norb, nelec, na, nb, seed, hf)
    rng = np.random.default_rng(seed)
    with _capi.Context(h1, eri, lib=lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert (ctx.na, ctx.nb, ctx.nelec) == (len(sa), len(sb), tuple(nelec))
        check_link_tables(ctx, sa, sb, norb, h1, eri)
        H, S2 = check_operators(ctx, h1, eri, sa, sb, norb, nelec, rng)
        check_ground_state(ctx, H, S2, h1, eri, sa, sb, norb, with_rdm2=with_rdm2, variants=variants)


def run_operator_parity(lib, norb, nelec, na, nb, seed, hf=False):
    """The sigma-kernel layouts only (H, S^2 and both penalty forms against the oracle): what the forced-layout
    tests need for their second, larger case -- the emulator pays for every barrier of a Davidson run."""
    h1, eri, sa, sb = make_problem(norb, nelec, na, nb, seed, hf)
    rng = np.random.default_rng(seed)
    with _capi.Context(h1, eri, lib=lib) as ctx:
        ctx.set_subspace(sa, sb)
        check_operators(ctx, h1, eri, sa, sb, norb, nelec, rng)
