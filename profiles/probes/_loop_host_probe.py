"""CPU probe: the HOST side of the SQD loop alone at config-3 shape (`_loop_pyprof.py`'s workload) behind a stub
``sci_solver`` that returns smooth random amplitudes: what `diagonalize_fermionic_hamiltonian` costs outside the solver."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S
from qiskit_addon_sqd_amd import sqd, fermion
norb, ne, nshots = 30, 8, 100_000
h1, eri = S.synthetic_integrals(norb)
rng = np.random.default_rng(7)
pool_a = np.sort(S.hf_centred_strings(norb, ne, 4000, 3)); pool_b = np.sort(S.hf_centred_strings(norb, ne, 4000, 5))
ia = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_a) - 1)
ib = np.minimum(rng.exponential(300.0, nshots).astype(int), len(pool_b) - 1)
def to_bits(x):
    return ((np.asarray(x, dtype=np.uint64)[:, None] >> np.arange(norb - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(bool)
bits = np.concatenate([to_bits(pool_b[ib]), to_bits(pool_a[ia])], axis=1)
bits ^= rng.random(bits.shape) < 0.02
t_solver = [0.0]; calls = [0]
def stub(ci_strings, one, two, norb_, nelec):
    t0 = time.perf_counter()
    out = []
    calls[0] += 1
    r = np.random.default_rng(1)
    for k, (sa, sb) in enumerate(ci_strings):
        wa = np.exp(-np.arange(len(sa)) / 40.0); wb = np.exp(-np.arange(len(sb)) / 40.0)
        amps = np.outer(r.permutation(wa), r.permutation(wb)); amps /= np.sqrt((amps * amps).sum())
        occ = (np.linspace(1, 0, norb_) ** 2 * ne / (np.linspace(1, 0, norb_) ** 2).sum()).clip(0, 1)
        st = fermion.SCIState(amps, sa, sb, norb_, nelec)
        out.append(fermion.SCIResult(-1.0 - 0.01 * k - 0.1 * calls[0], st, (occ, occ.copy())))
    t_solver[0] += time.perf_counter() - t0
    return out
def run():
    t_solver[0] = 0.0
    return sqd.diagonalize_fermionic_hamiltonian(h1, eri, bits, samples_per_batch=250, norb=norb, nelec=(ne, ne), num_batches=8,
                                                 max_iterations=4, seed=11, sci_solver=stub)
for _ in range(2): run()
t = time.perf_counter(); run(); tot = time.perf_counter() - t
print(f"run: {1e3*tot:.1f} ms, of which stub {1e3*t_solver[0]:.1f} ms -> host side {1e3*(tot - t_solver[0]):.1f} ms")
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue())
