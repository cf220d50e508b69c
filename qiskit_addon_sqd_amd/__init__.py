"""Importable name of the package whose sources live in ``qiskit-addon-sqd_amd/``.

The repository layout names the package directory ``qiskit-addon-sqd_amd`` (not a valid Python
identifier); this package has no modules of its own: its ``__path__`` points at that directory, so
``import qiskit_addon_sqd_amd`` / ``from qiskit_addon_sqd_amd.fermion import solve_fermion`` load the
sources from there (package documentation: ``qiskit-addon-sqd_amd/__init__.py``).
"""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "qiskit-addon-sqd_amd")
if not _os.path.isdir(_SRC):
    raise ImportError(f"qiskit_addon_sqd_amd: source directory {_SRC} not found")
__path__.insert(0, _SRC)

from ._version import __version__  # noqa: E402,F401
