import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
for na, nb in ((1000,1000),(2000,3000),(3000,3000),(900,4097)):
    sa, sb = S.hf_centred_strings(30, 8, na, 11), S.hf_centred_strings(30, 8, nb, 13)
    with _capi.Context(h1, eri) as ctx:
        ctx.set_subspace(sa, sb)
        ctx.time_sigma(1); ctx.time_sigma(1, 1, 0.0, 0.2)
        print(na, nb, ctx.sigma_kernel(), "H %.1f us | H + shift(S^2-ss) %.1f us | squared form %.1f us" % (1e3*ctx.time_sigma(3), 1e3*ctx.time_sigma(3, 1, 0.0, 0.2), 1e3*ctx.time_sigma(3, 2, 2.0, 0.2)), flush=True)
