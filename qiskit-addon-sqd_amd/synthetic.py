"""Seeded synthetic inputs for benchmarks and examples: integrals, FCIDUMP I/O and
particle-conserving bitstring generators.  No qiskit dependency (the reference's generators in
``counts.py:64-173`` return qiskit ``BitArray`` / count dictionaries; here plain bool matrices)."""

from __future__ import annotations

import numpy as np


def synthetic_integrals(norb: int, seed: int | None = None) -> tuple[np.ndarray, np.ndarray]:
    """Gapped one-body matrix and an 8-fold symmetric, positive-semidefinite ``eri`` built by density
    fitting ``eri = sum_L B_L (x) B_L`` (naux = 4 norb).  Seed default ``20260828 + norb``."""
    rng = np.random.default_rng(20260828 + norb if seed is None else seed)
    eps = -2.0 + 0.15 * np.arange(norb)
    m = rng.standard_normal((norb, norb))
    h1 = np.diag(eps) + 0.05 * 0.5 * (m + m.T)
    naux = 4 * norb
    B = rng.standard_normal((naux, norb, norb)) * (0.3 / np.sqrt(naux))
    B = 0.5 * (B + B.transpose(0, 2, 1))
    B[0] += 0.5 * np.eye(norb)
    eri = np.einsum("Lpq,Lrs->pqrs", B, B)
    return h1, eri


def write_fcidump(path, h1: np.ndarray, eri: np.ndarray, nelec: int, ms2: int = 0, ecore: float = 0.0, tol=1e-15):
    """Standard FCIDUMP text: ``&FCI NORB=,NELEC=,MS2=`` header, ``value i j k l`` lines, 1-based,
    chemist order, unique 8-fold entries only."""
    norb = h1.shape[0]
    with open(path, "w") as f:
        f.write(f" &FCI NORB={norb},NELEC={nelec},MS2={ms2},\n  ORBSYM={'1,' * norb}\n  ISYM=1,\n &END\n")
        for i in range(norb):
            for j in range(i + 1):
                ij = i * (i + 1) // 2 + j
                for k in range(norb):
                    for l in range(k + 1):
                        if k * (k + 1) // 2 + l > ij:
                            continue
                        v = eri[i, j, k, l]
                        if abs(v) > tol:
                            f.write(f"{v:23.16e} {i + 1} {j + 1} {k + 1} {l + 1}\n")
        for i in range(norb):
            for j in range(i + 1):
                if abs(h1[i, j]) > tol:
                    f.write(f"{h1[i, j]:23.16e} {i + 1} {j + 1} 0 0\n")
        f.write(f"{ecore:23.16e} 0 0 0 0\n")


def read_fcidump(path):
    """Returns ``(h1, eri, nelec, ms2, ecore)`` with the 8-fold symmetry expanded."""
    with open(path) as f:
        text = f.read()
    head, _, body = text.partition("&END")
    if not body:
        head, _, body = text.partition("/")
    import re

    norb = int(re.search(r"NORB\s*=\s*(\d+)", head).group(1))
    nelec = int(re.search(r"NELEC\s*=\s*(\d+)", head).group(1))
    m = re.search(r"MS2\s*=\s*(-?\d+)", head)
    ms2 = int(m.group(1)) if m else 0
    h1 = np.zeros((norb, norb))
    eri = np.zeros((norb,) * 4)
    ecore = 0.0
    for line in body.strip().splitlines():
        parts = line.split()
        if len(parts) != 5:
            continue
        v = float(parts[0].replace("D", "E"))
        i, j, k, l = (int(x) - 1 for x in parts[1:])
        if i < 0:
            ecore = v
        elif k < 0:
            h1[i, j] = h1[j, i] = v
        else:
            for a, b, c, d in ((i, j, k, l), (j, i, k, l), (i, j, l, k), (j, i, l, k),
                               (k, l, i, j), (l, k, i, j), (k, l, j, i), (l, k, j, i)):  # fmt: skip
                eri[a, b, c, d] = v
    return h1, eri, nelec, ms2, ecore


def _strings_to_half(strs: np.ndarray, norb: int) -> np.ndarray:
    """Integer strings -> bool matrix [n, norb], column 0 = most significant bit (reference layout)."""
    shifts = np.arange(norb - 1, -1, -1, dtype=np.uint64)
    return ((strs.astype(np.uint64)[:, None] >> shifts[None, :]) & np.uint64(1)).astype(bool)


def uniform_strings(norb: int, nelec: int, n: int, rng) -> np.ndarray:
    """``n`` distinct uniform-random particle-conserving strings, sorted ascending (int64)."""
    rng = np.random.default_rng(rng)
    out = np.zeros(0, dtype=np.uint64)
    while out.size < n:
        m = max(2 * (n - out.size), 64)
        pos = np.argsort(rng.random((m, norb)), axis=1)[:, :nelec].astype(np.uint64)
        new = np.bitwise_or.reduce(np.uint64(1) << pos, axis=1) if nelec else np.zeros(m, dtype=np.uint64)
        merged = np.concatenate([out, new])
        _, first = np.unique(merged, return_index=True)
        out = merged[np.sort(first)][:n]
    return np.sort(out).astype(np.int64)


def hf_centred_strings(norb: int, nelec: int, n: int, rng) -> np.ndarray:
    """``n`` distinct strings whose excitation rank from the aufbau string is Geometric(0.5)
    (capped), sorted ascending -- a well-connected subspace, unlike uniform sampling."""
    rng = np.random.default_rng(rng)
    hf = (1 << nelec) - 1
    out = {hf}
    kmax = min(nelec, norb - nelec)
    while len(out) < n:
        k = min(int(rng.geometric(0.5)), kmax)
        occ = rng.choice(nelec, k, replace=False)
        vir = nelec + rng.choice(norb - nelec, k, replace=False)
        s = hf
        for o in occ:
            s ^= 1 << int(o)
        for v in vir:
            s |= 1 << int(v)
        out.add(s)
    return np.array(sorted(out), dtype=np.int64)


def bitstring_matrix_from_strings(strs_a: np.ndarray, strs_b: np.ndarray, norb: int) -> np.ndarray:
    """Row i = [beta_i | alpha_i] as a bool matrix [n, 2 norb] (left half spin-down, right half spin-up,
    reference ``fermion.py:764-766``).  ``strs_a`` and ``strs_b`` must have equal length."""
    return np.concatenate([_strings_to_half(np.asarray(strs_b), norb), _strings_to_half(np.asarray(strs_a), norb)], 1)
