"""CPU oracle for the fermionic subspace projection + diagonalization path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import, link or execute it, and only as the checker.  The
product (``qiskit_addon_sqd_amd``) never imports this package and fails loudly
when the HIP library is missing.

PARITY STATUS: the integer layer (bitstring -> CI-string conversion, Hamming
checks) restates reference code that is importable (under stubs) in the
authoring container and is pinned by the reference's own literals and by
fixtures in ``tests/golden``.  The floating-point layer (sigma = Hc, Davidson,
RDMs, S^2) lives in third-party pyscf (``pyscf>=2.9``, reference
``pyproject.toml:30``) which is absent from ``/root/reference`` and from this
image: **parity unpinned** against pyscf itself.  It is anchored instead on an
independent brute-force second-quantised construction (Jordan-Wigner matrices,
``sqd_oracle.jw_*``) that every other oracle function and the HIP path are
checked against.

Round-4 search for an obtainable pyscf: ``import pyscf``
fails; ``pip download pyscf`` has no index; no pyscf wheel in ``/opt/wheelhouse``
(86 wheels), none under ``/opt/conda`` (pkgs / envs), no ``*pyscf*`` path anywhere
on this image's filesystem outside the reference's own text.  The pin therefore
stays as described above; ``tests/test_pyscf_opportunistic.py`` runs the pyscf
cross-check wherever ``import pyscf`` does succeed.
"""
