"""GPU probe: cProfile of the SQD loop at config-3 shape (profiles/probes/_loop_probe2.py's workload): where the host side
of diagonalize_fermionic_hamiltonian spends its time."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import sqd, fermion
norb, ne, nshots = 30, 8, 100_000
h1, eri = fermion.freeze_integrals(*S.synthetic_integrals(norb))
rng = np.random.default_rng(7)
pool_a = np.sort(S.hf_centred_strings(norb, ne, 4000, 3)); pool_b = np.sort(S.hf_centred_strings(norb, ne, 4000, 5))
ia = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_a) - 1)
ib = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_b) - 1)
def to_bits(x):
    return ((np.asarray(x, dtype=np.uint64)[:, None] >> np.arange(norb - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(bool)
bits = np.concatenate([to_bits(pool_b[ib]), to_bits(pool_a[ia])], axis=1)
bits ^= rng.random(bits.shape) < 0.02
def run():
    return sqd.diagonalize_fermionic_hamiltonian(h1, eri, bits, samples_per_batch=250, norb=norb, nelec=(ne, ne), num_batches=8,
                                                 max_iterations=4, seed=11)
for _ in range(2): run()
t = time.perf_counter(); run(); print(f"run: {1e3*(time.perf_counter()-t):.1f} ms")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue())
