import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F
h1, eri = F.freeze_integrals(*S.synthetic_integrals(30))
sa, sb = S.hf_centred_strings(30, 8, 317, 1001), S.hf_centred_strings(30, 8, 317, 1001 + 7919)
r = F.solve_sci((sa, sb), h1, eri, 30, (8, 8), compute_rdms=False)
print(r.energy)
