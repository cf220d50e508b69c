"""GPU tuning probe (not a test): sigma time of several builds of the library (tests/_variant_*.so)."""
import ctypes, glob, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from qiskit_addon_sqd_amd import _capi, synthetic as S
h1, eri = S.synthetic_integrals(30)
cases = (('hf', S.hf_centred_strings, 317, 317), ('uniform', S.uniform_strings, 317, 317), ('hf', S.hf_centred_strings, 1000, 1000),
         ('uniform', S.uniform_strings, 4000, 4000), ('hf', S.hf_centred_strings, 20000, 317), ('hf', S.hf_centred_strings, 707, 707))
libs = sorted(glob.glob(os.path.join(os.path.dirname(__file__), '_variant_*.so'))) + [str(_capi.LIB_PATH)]
for path in libs:
    lib = _capi.bind(ctypes.CDLL(path))
    row = []
    for name, gen, na, nb in cases:
        sa, sb = gen(30, 8, na, 1001), gen(30, 8, nb, 1001 + 7919)
        with _capi.Context(h1, eri, lib=lib) as ctx:
            ctx.set_subspace(sa, sb)
            ctx.time_sigma(3)
            row.append(f"{name[:2]}{na}x{nb}={ctx.time_sigma(10) * 1e3:8.1f}")
    print(os.path.basename(path), ' '.join(row), flush=True)
