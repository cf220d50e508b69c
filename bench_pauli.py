#!/usr/bin/env python
"""Secondary benchmark (not the driver's bench.py): the qubit / Pauli projection path.

(1) the reference's own published benchmark shape (docs/guides/benchmark_pauli_projection.ipynb:
    matrix_elements_from_pauli(Z^(x)40) on d sorted unique 40-bit strings; 4.173 s at d = 49 998 839 on
    unspecified hardware => ~12 M rows/s per term), and
(2) BASELINE config 5: 40-qubit random 2-local Hamiltonian projected onto 1e5 bitstrings.
Prints one JSON line per case; the numpy oracle (CPU restatement of qubit.py) is timed on a bounded
sample of the same workload on this box's host."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def unique_rows(nq, d, rng):
    vals = np.unique(rng.integers(0, 1 << nq, size=int(d * 1.02), dtype=np.uint64))[:d]
    shifts = np.arange(nq - 1, -1, -1, dtype=np.uint64)
    return ((vals[:, None] >> shifts[None, :]) & np.uint64(1)).astype(bool), vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=5_000_000, help="rows for the Z^40 case (reference used 5e7)")
    ap.add_argument("--d2", type=int, default=100_000, help="rows for the 2-local case")
    ap.add_argument("--skip-cpu", action="store_true")
    args = ap.parse_args()
    from qiskit_addon_sqd_amd import qubit as Q

    rng = np.random.default_rng(7)
    nq = 40
    # ---- (1) single diagonal term
    mat, vals = unique_rows(nq, args.d, rng)
    term = Q.PauliTerm.from_label("Z" * nq)
    Q.matrix_elements_from_pauli(mat[:1000], term)  # warm-up (library load, context)
    rows_u64 = Q._rows_to_uint64(mat)
    t0 = time.perf_counter()
    indptr, indices, data, ms_k = Q._project(rows_u64, [(0, [((1 << nq) - 1, 1.0)])])
    t_native = time.perf_counter() - t0
    t0 = time.perf_counter()
    amp, r, c = Q.matrix_elements_from_pauli(mat, term)
    t_api = time.perf_counter() - t0
    out = {"case": "matrix_elements_from_pauli(Z^40)", "rows": int(mat.shape[0]), "kernel_ms": ms_k,
           "native_call_s": t_native, "python_api_s": t_api, "rows_per_s_kernel": mat.shape[0] / (ms_k * 1e-3),
           "rows_per_s_native_call": mat.shape[0] / t_native, "rows_per_s_python_api": mat.shape[0] / t_api,
           "reference_published_rows_per_s": 49998839 / 4.173,
           "algorithmic_bytes_per_row": 32, "kernel_GBs": 32 * mat.shape[0] / (ms_k * 1e-3) / 1e9}
    if not args.skip_cpu:
        from oracle import qubit_oracle as QO

        n = min(mat.shape[0], 200_000)
        t0 = time.perf_counter()
        QO.matrix_elements_from_pauli(mat[:n], term.x, term.z)
        out["cpu_oracle_rows_per_s"] = n / (time.perf_counter() - t0)
        out["cpu_oracle_sample_rows"] = n
    print(json.dumps(out), flush=True)

    # ---- (2) random 2-local Hamiltonian, 40 qubits
    mat2, _ = unique_rows(nq, args.d2, rng)
    # make the subspace connected: add partners under random 1- and 2-qubit flips
    flips = np.zeros((400, nq), dtype=bool)
    for f in flips:
        f[rng.choice(nq, rng.integers(1, 3), replace=False)] = True
    extra = mat2[rng.integers(mat2.shape[0], size=args.d2 // 2)] ^ flips[rng.integers(400, size=args.d2 // 2)]
    mat2 = Q.sort_and_remove_duplicates(np.concatenate([mat2, extra]))[: args.d2]
    labels, coeffs = [], []
    for i in range(nq):
        for j in range(i + 1, nq):
            for a in "XYZ":
                for b in "XYZ":
                    lab = ["I"] * nq
                    lab[i], lab[j] = a, b
                    labels.append("".join(lab)); coeffs.append(float(rng.standard_normal()))
    for i in range(nq):
        for a in "XYZ":
            lab = ["I"] * nq
            lab[i] = a
            labels.append("".join(lab)); coeffs.append(float(rng.standard_normal()))
    ham = Q.PauliSum.from_list(list(zip(labels, coeffs)))
    t0 = time.perf_counter()
    op = Q.project_operator_to_subspace(mat2, ham)
    t_api = time.perf_counter() - t0
    out2 = {"case": "project_operator_to_subspace(random 2-local, 40 qubits)", "rows": int(mat2.shape[0]),
            "terms": len(labels), "nnz": int(op.nnz), "python_api_s": t_api,
            "row_terms_per_s": mat2.shape[0] * len(labels) / t_api}
    if not args.skip_cpu:
        from oracle import qubit_oracle as QO

        nt = 40
        t0 = time.perf_counter()
        for l in labels[:nt]:
            t = Q.PauliTerm.from_label(l)
            QO.matrix_elements_from_pauli(mat2, t.x, t.z)
        dt = time.perf_counter() - t0
        out2["cpu_oracle_row_terms_per_s"] = mat2.shape[0] * nt / dt
        out2["cpu_oracle_sample_terms"] = nt
        out2["cpu_oracle_est_total_s"] = dt / nt * len(labels)
    print(json.dumps(out2), flush=True)


if __name__ == "__main__":
    main()
