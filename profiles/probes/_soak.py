"""GPU soak (not a test): random small problems through every sigma kernel (forced by the env hooks) against the dense
oracle -- sigma, S^2, both penalty forms, Davidson energy against the oracle's pyscf-flow Davidson.  Prints failures; exit code 1 if any."""
import os, sys, itertools
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np
from math import comb
from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi

lib = _capi.load_library()
rng = np.random.default_rng(int(os.environ.get('SOAK_SEED', '1')))
ncase = int(os.environ.get('SOAK_CASES', '200'))
fails = 0
modes = [{}, {'SQD_SIGMA_DIRECT': '1'}, {'SQD_SIGMA_DIRECT': '0'}, {'SQD_SIGMA_ROWS': '1'}, {'SQD_SIGMA_ROWS': '2'},
         {'SQD_SIGMA_ROWS': '3'}, {'SQD_SIGMA_ROWS': '8'}, {'SQD_ELL_CAP': '2', 'SQD_SIGMA_DIRECT': '0'},
         {'SQD_SIGMA_GLOBAL_ROWS': '64', 'SQD_SIGMA_DIRECT': '0'}, {'SQD_SIGMA_PASS': '5', 'SQD_SIGMA_DIRECT': '0'},
         {'SQD_SIGMA_L': '2', 'SQD_SIGMA_DIRECT': '0'}, {'SQD_SIGMA_L0': '0', 'SQD_SIGMA_DIRECT': '0'}]
BIG = int(os.environ.get('SOAK_BIG', '0'))
for case in range(ncase):
    norb = int(rng.integers(3, 14 if BIG else 11))
    ne = (int(rng.integers(1, norb)), int(rng.integers(1, norb)))
    na = int(rng.integers(1, min(comb(norb, ne[0]), 130 if BIG else 70) + 1))
    nb = int(rng.integers(1, min(comb(norb, ne[1]), 260 if BIG else 140) + 1))
    while na * nb > 9000:
        na = max(1, na // 2)
    hf = bool(rng.integers(0, 2))
    seed = int(rng.integers(0, 10**6))
    h1, eri = O.synthetic_integrals(norb, seed=seed)
    gen = O.hf_centred_strings if hf else O.random_strings
    sa, sb = gen(norb, ne[0], na, seed + 1), gen(norb, ne[1], nb, seed + 2)
    H = O.build_php(h1, eri, sa, sb, norb)
    S2 = O.build_spin_square(sa, sb, norb, ne)
    x = rng.standard_normal((na, nb))
    w = np.linalg.eigvalsh(H)
    ss = float(rng.choice([0.0, 0.75, 2.0]))
    mode = modes[case % len(modes)]
    for k in ('SQD_SIGMA_DIRECT', 'SQD_SIGMA_ROWS', 'SQD_ELL_CAP', 'SQD_SIGMA_GLOBAL_ROWS', 'SQD_SIGMA_PASS', 'SQD_SIGMA_L', 'SQD_SIGMA_L0'):
        os.environ.pop(k, None)
    os.environ.update(mode)
    try:
        with _capi.Context(h1, eri, lib=lib) as ctx:
            ctx.set_subspace(sa, sb)
            kern = ctx.sigma_kernel()
            scale = max(1.0, np.abs(H).max()) * np.sqrt(na * nb)
            errs = {
                'sigma': np.abs(ctx.sigma(x).ravel() - H @ x.ravel()).max(),
                's2': np.abs(ctx.contract_ss(x).ravel() - S2 @ x.ravel()).max(),
                'pen1': np.abs(ctx.sigma(x, 1, ss, 0.3).ravel() - (H @ x.ravel() + 0.3 * (S2 @ x.ravel() - ss * x.ravel()))).max(),
            }
            P = S2 - ss * np.eye(na * nb)
            errs['pen2'] = np.abs(ctx.sigma(x, 2, ss, 0.3).ravel() - (H @ x.ravel() + 0.3 * (P @ (P @ x.ravel())))).max()
            amps, st = ctx.davidson(max_cycle=300)
            # the reference flow from pyscf's start vector (it may lack overlap with the ground state by symmetry: then
            # pyscf, the oracle and this library all converge to the same excited state)
            x0 = O.init_guess(np.diag(H), na, nb, nelec=ne)
            _, e_flow, _, _ = O.davidson_pyscf(lambda y: H @ y, x0 / np.linalg.norm(x0), np.diag(H), tol=1e-11, max_cycle=300)
            bad = {k: v for k, v in errs.items() if not (v < 1e-10 * scale)}
            if not st['converged'] or abs(st['e_davidson'] - e_flow) > 1e-8:
                bad['e0'] = (st['e_davidson'], e_flow, w[0], st['converged'])
            if bad:
                fails += 1
                print('FAIL', case, dict(norb=norb, ne=ne, na=na, nb=nb, hf=hf, seed=seed, ss=ss), mode, kern, bad, flush=True)
    except Exception as exc:
        fails += 1
        print('EXC ', case, dict(norb=norb, ne=ne, na=na, nb=nb, hf=hf, seed=seed), mode, repr(exc), flush=True)
print(f'soak: {ncase} cases, {fails} failures')
sys.exit(1 if fails else 0)
