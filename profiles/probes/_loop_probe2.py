"""GPU probe (not a test): the subspaces a real SQD run produces (configuration recovery + carry-over; reference
fermion.py:563-640) -- BASELINE config 3 shape: N2-sized (16e,30o), 1e5 sampled bitstrings, 8 subsample batches per
iteration.  Per iteration and batch: dimensions, link densities of both spins, the sigma kernel selected, sigma builds;
per iteration the wall clock of the batched solve.  env DUMP=path: the CI strings of the last iteration as .npz."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import sqd, fermion

norb, ne, nshots = 30, 8, 100_000
h1, eri = fermion.freeze_integrals(*S.synthetic_integrals(norb))
rng = np.random.default_rng(7)
pool_a = np.sort(S.hf_centred_strings(norb, ne, 4000, 3))
pool_b = np.sort(S.hf_centred_strings(norb, ne, 4000, 5))
ia = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_a) - 1)
ib = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_b) - 1)


def to_bits(x):
    return ((np.asarray(x, dtype=np.uint64)[:, None] >> np.arange(norb - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(bool)


bits = np.concatenate([to_bits(pool_b[ib]), to_bits(pool_a[ia])], axis=1)
bits ^= rng.random(bits.shape) < 0.02
mode = os.environ.get('MODE', 'batched')
log = []


def timed_solver(ci_strings, h, g, norb, nelec):
    kw = {} if mode == 'batched' else {'concurrency': int(mode)}
    t0 = time.perf_counter()
    out = fermion.solve_sci_batch(ci_strings, h, g, norb, nelec, spin_sq=0.0, **kw)
    dt = time.perf_counter() - t0
    rows = []
    if mode == 'batched':
        ctx = fermion._get_context(h, g, 0, slot='batch')
        for i, (a, b) in enumerate(ci_strings):
            sub = ctx.batch_sub(i)
            (sa_, da_), (sb_, db_) = sub.link_counts(0), sub.link_counts(1)
            st = fermion._TLS.batch_stats[i]
            rows.append((len(a), len(b), sa_, da_, sb_, db_, sub.sigma_kernel(), st['n_sigma'], sub.sigma_bytes()))
    log.append((dt, rows, ci_strings))
    return out


for rep in range(2):
    log.clear()
    t0 = time.perf_counter()
    res = sqd.diagonalize_fermionic_hamiltonian(h1, eri, bits, samples_per_batch=250, norb=norb, nelec=(ne, ne),
                                                num_batches=8, max_iterations=4, sci_solver=timed_solver, seed=11)
    t = time.perf_counter() - t0
print(f"mode {mode}: total {t*1e3:.1f} ms for {len(log)} iterations; energy {res.energy:.6f}")
for i, (dt, rows, _) in enumerate(log):
    print(f"  iteration {i}: solver {dt*1e3:7.2f} ms for {len(rows) or 8} batches")
    for r in rows:
        na, nb, sa_, da_, sb_, db_, kern, nsig, bsig = r
        print(f"     {na:5d} x {nb:5d} D={na*nb:8d}  alpha links/string s {sa_/na:6.1f} d {da_/na:7.1f} (same-spin density {(sa_+da_)/na/na*100:5.1f} %)"
              f"  beta s {sb_/nb:6.1f} d {db_/nb:7.1f} ({(sb_+db_)/nb/nb*100:5.1f} %)  {kern:16s} {nsig:3d} sigma builds, B_sigma {bsig/1e6:7.2f} MB")
print(f"  host-side sample processing (everything else): {(t - sum(x[0] for x in log))*1e3:.1f} ms")
if os.environ.get('DUMP'):
    cs = log[-1][2]
    np.savez(os.environ['DUMP'], **{f'a{i}': a for i, (a, b) in enumerate(cs)}, **{f'b{i}': b for i, (a, b) in enumerate(cs)})
