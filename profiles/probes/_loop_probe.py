"""GPU probe (not a test): one full SQD run (configuration recovery loop) on synthetic samples -- where does an
iteration's wall-clock go (host-side sample processing vs the batched solves)?  BASELINE config 3 shape:
N2-sized (16e,30o), 1e5 sampled bitstrings, 8 subsample batches per iteration.  env CONC = batches in flight."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import sqd, fermion

norb, ne, nshots = 30, 8, 100_000
h1, eri = S.synthetic_integrals(norb)
rng = np.random.default_rng(7)
# noisy HF-centred samples: half-strings from an HF-centred pool (low ranks favoured), 2 % bit-flip noise
pool_a = np.sort(S.hf_centred_strings(norb, ne, 4000, 3))
pool_b = np.sort(S.hf_centred_strings(norb, ne, 4000, 5))
ia = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_a) - 1)
ib = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_b) - 1)


def to_bits(x):
    return ((np.asarray(x, dtype=np.uint64)[:, None] >> np.arange(norb - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(bool)


bits = np.concatenate([to_bits(pool_b[ib]), to_bits(pool_a[ia])], axis=1)  # left half = beta, right half = alpha
bits ^= rng.random(bits.shape) < 0.02
conc = int(os.environ.get('CONC', '1'))
solve_t = []


def timed_solver(ci_strings, h, g, norb, nelec):
    t0 = time.perf_counter()
    out = fermion.solve_sci_batch(ci_strings, h, g, norb, nelec, spin_sq=0.0, concurrency=conc)
    solve_t.append((time.perf_counter() - t0, [len(a) * len(b) for a, b in ci_strings]))
    return out


for rep in range(2):  # second repetition: arenas and contexts are warm
    solve_t.clear()
    t0 = time.perf_counter()
    res = sqd.diagonalize_fermionic_hamiltonian(h1, eri, bits, samples_per_batch=250, norb=norb, nelec=(ne, ne),
                                                num_batches=8, max_iterations=4, sci_solver=timed_solver, seed=11)
    t = time.perf_counter() - t0
print(f"concurrency {conc}: total {t*1e3:.1f} ms for {len(solve_t)} iterations; energy {res.energy:.6f}")
for i, (ts, dims) in enumerate(solve_t):
    print(f"  iteration {i}: solver {ts*1e3:7.2f} ms for 8 batches, D = {dims}")
print(f"  host-side sample processing (everything else): {(t - sum(x[0] for x in solve_t))*1e3:.1f} ms")
