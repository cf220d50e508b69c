#!/usr/bin/env python
"""Generate golden input/output vectors by importing the REFERENCE (qiskit-addon-sqd, read-only at
/root/reference) in the authoring container, with stub modules for its absent third-party imports
(qiskit, jax, pyscf).  Only DATA (inputs and the reference's outputs) is written to
tests/golden/*.npz|json; no reference source travels.  Run:  python tests/golden/make_golden.py

Covers every reference function on the hot path's *integer* boundary that is executable here
(SURVEY.md 8c): counts.bitstring_matrix_to_integers, fermion.bitstring_matrix_to_ci_strs,
fermion._check_ci_strs, plus the string-preparation seam of the SQD loop (_prepare_ci_strings with
postselect/subsample underneath) whose sorted int64 string pairs are what ``sci_solver`` receives.
The floating-point arithmetic lives in pyscf (absent): parity unpinned there, see oracle/__init__.py.
"""
import json
import sys
import types
from pathlib import Path

import numpy as np

REF = "/root/reference"
OUT = Path(__file__).resolve().parent


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def install_stubs():
    class BitArray:  # stand-in for qiskit.primitives.BitArray (array + num_bits + num_shots)
        def __init__(self, array, num_bits):
            self.array, self.num_bits = array, num_bits

        num_shots = property(lambda self: self.array.shape[0])

        @classmethod
        def from_bool_array(cls, b):
            b = np.asarray(b, bool)
            pad = (-b.shape[-1]) % 8
            full = np.concatenate([np.zeros(b.shape[:-1] + (pad,), bool), b], -1)
            return cls(np.packbits(full, -1), b.shape[-1])

    _mod("qiskit")
    _mod("qiskit.primitives", BitArray=BitArray)
    _mod("qiskit.utils")
    _mod("qiskit.utils.deprecation", deprecate_func=lambda **kw: (lambda f: f))
    _mod("qiskit.quantum_info", Pauli=object, SparsePauliOp=object)
    cfg = types.SimpleNamespace(update=lambda *a: None)
    jnp = _mod("jax.numpy")
    jsl = _mod("jax.scipy.linalg", expm=None)
    _mod("jax.scipy", linalg=jsl)
    _mod("jax", Array=object, config=cfg, grad=lambda f, **k: f, jit=lambda f: f, vmap=lambda f, *a, **k: f, numpy=jnp)
    names = ["_as_SCIvector", "make_rdm1", "make_rdm1s", "make_rdm2", "make_rdm2s", "spin_square"]
    sel = _mod("pyscf.fci.selected_ci", **{n: None for n in names})
    _mod("pyscf", fci=_mod("pyscf.fci", selected_ci=sel))
    return BitArray


def main():
    BitArray = install_stubs()
    sys.path.insert(0, REF)
    import qiskit_addon_sqd.counts as RC
    import qiskit_addon_sqd.fermion as RF

    rng = np.random.default_rng(20260928)
    cases = {}

    # --- bitstring_matrix_to_integers (counts.py:186-201): widths below / at / above 64 bits
    for nbits in (1, 8, 30, 57, 63, 64, 70):
        mat = rng.integers(2, size=(11, nbits)).astype(bool)
        out = RC.bitstring_matrix_to_integers(mat)
        cases[f"b2i_{nbits}"] = dict(matrix=mat.astype(np.uint8).tolist(), out=[str(int(x)) for x in out],
                                     dtype=str(out.dtype))

    # --- bitstring_matrix_to_ci_strs (fermion.py:1004-1035): literals of the docs + random sets
    lit = np.array([[0, 0, 0, 1, 0, 0, 1, 0], [0, 1, 0, 0, 1, 0, 0, 0]], dtype=bool)
    for open_shell in (False, True):
        a, b = RF.bitstring_matrix_to_ci_strs(lit, open_shell=open_shell)
        cases[f"ci_lit_open{int(open_shell)}"] = dict(matrix=lit.astype(np.uint8).tolist(), open_shell=open_shell,
                                                      a=[str(int(x)) for x in a], b=[str(int(x)) for x in b])
    for norb, n in ((6, 40), (30, 200), (32, 64)):
        mat = rng.integers(2, size=(n, 2 * norb)).astype(bool)
        for open_shell in (False, True):
            a, b = RF.bitstring_matrix_to_ci_strs(mat, open_shell=open_shell)
            cases[f"ci_{norb}_{n}_open{int(open_shell)}"] = dict(
                matrix=mat.astype(np.uint8).tolist(), open_shell=open_shell,
                a=[str(int(x)) for x in a], b=[str(int(x)) for x in b], dtype=str(np.asarray(a).dtype))

    # --- _check_ci_strs (fermion.py:1075-1097): valid input (sorted unique output) and both error texts
    a = np.array([0b0111, 0b1011, 0b1101, 0b1011], dtype=np.int64)
    b = np.array([0b0011, 0b0101], dtype=np.int64)
    oa, ob = RF._check_ci_strs((a, b))
    cases["check_ok"] = dict(a=a.tolist(), b=b.tolist(), out_a=oa.tolist(), out_b=ob.tolist())
    for key, (xa, xb) in {"check_bad_up": (np.array([7, 11, 3]), b), "check_bad_dn": (a, np.array([3, 5, 7]))}.items():
        try:
            RF._check_ci_strs((xa, xb))
            msg = None
        except ValueError as exc:
            msg = str(exc)
        cases[key] = dict(a=np.asarray(xa).tolist(), b=np.asarray(xb).tolist(), error=msg)

    # --- the seam: what ``sci_solver`` receives.  Run the reference's public SQD loop
    # (diagonalize_fermionic_hamiltonian, fermion.py:204-462) with a deterministic solver plug-in built on
    # the numpy oracle, recording every list of (strs_a, strs_b) handed across the seam (fermion.py:432)
    # and the final result.  Inputs + recorded outputs are the fixture.
    sys.path.insert(0, str(OUT.parent.parent))
    from oracle import sqd_oracle as O

    loops = {}
    for name, symm, max_dim, seed in (("loop_open", False, None, 1234), ("loop_symm", True, 12, 4321)):
        norb, nelec = 6, (3, 3)
        h1, eri = O.synthetic_integrals(norb, seed=5)
        nshots = 300
        half = lambda ne: np.array(
            [rng.permutation(np.r_[np.ones(ne, bool), np.zeros(norb - ne, bool)]) for _ in range(nshots)])
        bits = np.concatenate([half(nelec[1]), half(nelec[0])], axis=1)
        noisy = bits ^ (rng.random(bits.shape) < 0.03)
        calls = []

        def fake_solver(ci_strings, one, two, norb_, nelec_):
            out = []
            calls.append([(np.asarray(a).copy(), np.asarray(b).copy()) for a, b in ci_strings])
            for sa, sb in ci_strings:
                e, amps, occ, _, _ = O.solve_fermion_dense((sa, sb), one, two)
                out.append(RF.SCIResult(e, RF.SCIState(amps, sa, sb, norb_, nelec_), occ))
            return out

        res = RF.diagonalize_fermionic_hamiltonian(
            h1, eri, BitArray.from_bool_array(noisy), samples_per_batch=20, norb=norb, nelec=nelec,
            num_batches=3, max_iterations=4, symmetrize_spin=symm, max_dim=max_dim, sci_solver=fake_solver,
            carryover_threshold=1e-3, seed=seed)
        loops[name] = dict(
            norb=norb, nelec=list(nelec), integrals_seed=5, noisy=noisy.astype(np.uint8).tolist(),
            samples_per_batch=20, num_batches=3, max_iterations=4, symmetrize_spin=symm, max_dim=max_dim,
            carryover_threshold=1e-3, seed=seed,
            calls=[[dict(a=[int(x) for x in a], b=[int(x) for x in b], dtype=str(a.dtype)) for a, b in call] for call in calls],
            energy=float(res.energy), strs_a=[int(x) for x in res.sci_state.ci_strs_a],
            strs_b=[int(x) for x in res.sci_state.ci_strs_b],
            abs_amplitudes=np.abs(res.sci_state.amplitudes).tolist(),
            occ_a=np.asarray(res.orbital_occupancies[0]).tolist(), occ_b=np.asarray(res.orbital_occupancies[1]).tolist())
    (OUT / "sqd_loop.json").write_text(json.dumps(loops))
    print("wrote", OUT / "sqd_loop.json", [(k, len(v["calls"])) for k, v in loops.items()])

    (OUT / "integer_layer.json").write_text(json.dumps(cases))
    print("wrote", OUT / "integer_layer.json", "with", len(cases), "cases")


if __name__ == "__main__":
    main()
