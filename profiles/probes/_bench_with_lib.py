"""GPU probe (not a test): bench.py against another build of the library (A/B on one box):
   SQD_LIB=profiles/probes/_build/libsqd_hip_old.so python profiles/probes/_bench_with_lib.py <bench.py arguments>"""
import os, runpy, sys
from pathlib import Path
ROOT = Path(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
sys.path.insert(0, str(ROOT))
from qiskit_addon_sqd_amd import _capi
if os.environ.get('SQD_LIB'):
    _capi.LIB_PATH = ROOT / os.environ['SQD_LIB']
sys.argv = [str(ROOT / 'bench.py')] + sys.argv[1:]
runpy.run_path(str(ROOT / 'bench.py'), run_name='__main__')
