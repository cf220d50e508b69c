import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np
from _parity import make_problem
from qiskit_addon_sqd_amd import _capi
h1, eri, sa, sb = make_problem(8, (4, 4), 70, 70, 17, False)
with _capi.Context(h1, eri) as ctx:
    ctx.set_subspace(sa, sb)
    amps, st = ctx.davidson()
    print('main', st['converged'], st['iterations'], st['e_davidson'])
    rng = np.random.default_rng(amps.size)
    ci0 = 37.5 * (amps + 0.05 * rng.standard_normal(amps.shape))
    for mc in (100, 200, 400):
        a3, st3 = ctx.davidson(ci0, max_cycle=mc)
        print('ci0 max_cycle', mc, st3['converged'], st3['iterations'], st3['e_davidson'], st3.get('n_sigma'))
