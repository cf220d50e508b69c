"""GPU probe (not a test): where do the sporadic ~30-50 ms host stalls come from?  Repeats one HF-centred solve and
prints every call that took > 3x the median, with the cumulative number of kernel launches before it."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from qiskit_addon_sqd_amd import synthetic as S, fermion as F

h1, eri = S.synthetic_integrals(30)
ctx = F._get_context(h1, eri, 0)
kind = os.environ.get('KIND', 'hf')
gen = S.hf_centred_strings if kind == 'hf' else S.uniform_strings
sa, sb = gen(30, 8, 317, 1000), gen(30, 8, 317, 1000 + 7919)
n = int(os.environ.get('N', '150'))
ts, iters = [], []
for i in range(n):
    t0 = time.perf_counter()
    _, st, _ = ctx.solve(sa, sb)
    ts.append((time.perf_counter() - t0) * 1e3)
    iters.append(st['iterations'])
ts = np.array(ts)
launches = np.cumsum([6 + 4 * (it + 1) + 4 for it in iters])
med = np.median(ts)
out = [(i, round(float(ts[i]), 2), int(launches[i])) for i in np.nonzero(ts > 3 * med)[0]]
print(f"{kind} env={ {k: os.environ[k] for k in ('HSA_ENABLE_INTERRUPT', 'GPU_MAX_HW_QUEUES', 'HIP_FORCE_DEV_KERNARG', 'ROC_SIGNAL_POOL_SIZE', 'AMD_DIRECT_DISPATCH') if k in os.environ} } "
      f"median {med:.3f} ms, total {ts.sum():.1f} ms, outliers (call, ms, launches so far): {out}", flush=True)
