"""GPU probe (not a test): 20 sigma launches on one subspace (for rocprofv3 --pmc runs).  argv: hf|un n"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
kind, n = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ('hf', 317)
h1, eri = S.synthetic_integrals(30)
gen = S.hf_centred_strings if kind == 'hf' else S.uniform_strings
sa, sb = gen(30, 8, n, 1001), gen(30, 8, n, 1001 + 7919)
with _capi.Context(h1, eri) as ctx:
    ctx.set_subspace(sa, sb)
    print(kind, n, ctx.time_sigma(20) * 1e3, 'us per sigma')
