"""MI355X-native drop-in for the fermionic subspace-diagonalization path of qiskit-addon-sqd.

Public surface mirrors ``qiskit_addon_sqd.fermion`` for that path (reference
``qiskit_addon_sqd/fermion.py``): ``solve_fermion``, ``solve_sci``, ``solve_sci_batch``,
``SCIState``, ``SCIResult``, ``bitstring_matrix_to_ci_strs``.  All arithmetic runs in
``libsqd_hip.so`` (hand-written HIP for gfx950) behind the C ABI of ``include/sqd_hip.h``.

Modules: ``fermion`` (solver surface), ``sqd`` (the configuration-recovery loop
``diagonalize_fermionic_hamiltonian``), ``sampling`` (post-selection, subsampling, configuration
recovery, counts conversion), ``distributed`` (one-process-per-GPU batch-sharded ``sci_solver``),
``synthetic`` (seeded inputs, FCIDUMP I/O), ``_capi`` (ctypes binding).
"""
from ._version import __version__  # noqa: F401
