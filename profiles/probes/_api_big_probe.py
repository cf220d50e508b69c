"""GPU probe: the public entry point on connected sets of D = 1e6 ... 2.5e7 -- solve_sci (lazy RDMs, as the SQD loop calls
it), with spin_sq = 0, and with compute_rdms=True (rdm1 + rdm2 on the device): wall clock per call beside the Davidson
run's own time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
for n in (int(v) for v in os.environ.get("SIZES", "1000 3000 5000").split()):
    sa, sb = S.hf_centred_strings(30, 8, n, 11), S.hf_centred_strings(30, 8, n, 13)
    line = f"hf {n}^2:"
    for name, kw in (("lazy", {}), ("spin_sq=0", {"spin_sq": 0.0}), ("compute_rdms", {"compute_rdms": True})):
        F.solve_sci((sa, sb), h1, eri, 30, (8, 8), **kw)
        t0 = time.perf_counter()
        r = F.solve_sci((sa, sb), h1, eri, 30, (8, 8), **kw)
        amps = r.sci_state.amplitudes  # (the state on the host, as a caller reads it)
        if kw.get("compute_rdms"):
            _ = r.rdm2
        ms = 1e3 * (time.perf_counter() - t0)
        line += f"  {name}: {ms:8.1f} ms (E = {r.energy:.8f})"
    print(line, flush=True)
