"""Importable name of the package whose sources live in ``qiskit-addon-sqd_amd/``.

The repository layout names the package directory ``qiskit-addon-sqd_amd`` (not a valid Python
identifier); this shim points the import system at it so that
``import qiskit_addon_sqd_amd`` / ``from qiskit_addon_sqd_amd.fermion import solve_fermion`` work.
"""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "qiskit-addon-sqd_amd")
__path__.insert(0, _SRC)
with open(_os.path.join(_SRC, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_SRC, "__init__.py"), "exec"))
del _f
