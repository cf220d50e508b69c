"""ctypes front-end of oracle/sci_ref.c (O2: CPU restatement of pyscf's selected-CI algorithm).

TEST INFRASTRUCTURE ONLY -- checker and ``cpu_baseline`` of bench.py.  The orchestration in
``solve_fermion_ref`` follows reference ``qiskit_addon_sqd/fermion.py:745-845`` step by step, with
pyscf's pieces replaced by the C restatement.  Parity unpinned against pyscf.
"""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess
import time
from pathlib import Path

import numpy as np

from . import sqd_oracle as O

_HERE = Path(__file__).resolve().parent
_LIBP = _HERE / "_build" / "libsci_ref.so"
_lib = None
_blas_name = "internal loop"


def _find_openblas_dgemm():
    """cblas_dgemm of the OpenBLAS bundled with numpy/scipy (ILP64 'scipy_' prefixed build)."""
    global _blas_name
    import numpy
    import scipy

    cands = []
    for mod in (scipy, numpy):
        base = Path(mod.__file__).resolve().parent.parent
        cands += glob.glob(str(base / f"{mod.__name__}.libs" / "libscipy_openblas*.so"))
    for path in cands:
        try:
            lib = C.CDLL(path)
        except OSError:
            continue
        for sym, setter in (("scipy_cblas_dgemm64_", "scipy_openblas_set_num_threads64_"),):
            if hasattr(lib, sym):
                if hasattr(lib, setter):
                    getattr(lib, setter)(1)  # OpenMP threads outside, sequential BLAS inside (as pyscf)
                _blas_name = f"OpenBLAS ({os.path.basename(path)}, 1 thread per call)"
                return lib, C.cast(getattr(lib, sym), C.c_void_p)
    return None, None


def load(use_blas: bool = True):
    global _lib, _keep
    if _lib is not None:
        return _lib
    if not _LIBP.exists() or _LIBP.stat().st_mtime < (_HERE / "sci_ref.c").stat().st_mtime:
        subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    lib = C.CDLL(str(_LIBP))
    lib.ref_des_uniq_strs.restype = C.c_int64
    lib.ref_num_threads.restype = C.c_int
    if use_blas:
        blas, fn = _find_openblas_dgemm()
        if fn:
            lib.ref_set_dgemm(fn)
            lib._blas_keepalive = blas
    # physical cores, capped at 64: the OpenBLAS bundled with numpy/scipy keeps per-thread metadata for 64
    # callers and crashes (SIGSEGV) when dgemm is entered from more OpenMP threads than that -- which a
    # 256-thread GPU host does by default
    lib.ref_set_threads(max(1, min(64, (os.cpu_count() or 2) // 2)))
    _lib = lib
    return lib


def blas_name() -> str:
    return _blas_name


def num_threads() -> int:
    return int(load().ref_num_threads())


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class RefProblem:
    """Link tables, hdiag and absorbed integrals of one subspace, pyscf layout."""

    def __init__(self, h1, eri, strs_a, strs_b):
        lib = load()
        self.h1 = np.ascontiguousarray(h1, dtype=np.float64)
        self.norb = norb = self.h1.shape[0]
        self.eri = np.ascontiguousarray(eri, dtype=np.float64).reshape((norb,) * 4)
        self.sa = np.ascontiguousarray(np.asarray(strs_a).astype(np.uint64))
        self.sb = np.ascontiguousarray(np.asarray(strs_b).astype(np.uint64))
        self.na, self.nb = len(self.sa), len(self.sb)
        self.nelec = (bin(int(self.sa[0])).count("1"), bin(int(self.sb[0])).count("1"))
        t0 = time.perf_counter()
        self.cd, self.dd, self.nlink, self.ndl, self.ninter = [], [], [], [], []
        for strs, nocc in ((self.sa, self.nelec[0]), (self.sb, self.nelec[1])):
            n = len(strs)
            nvir = norb - nocc
            nlink = nocc + nocc * nvir
            cd = np.zeros((n, nlink, 4), dtype=np.int32)
            lib.ref_cre_des_linkstr_tril(_p(cd), norb, C.c_int64(n), nocc, _p(strs))
            npair = max(1, nocc * (nocc - 1) // 2)
            inter = np.zeros(n * npair, dtype=np.uint64)
            ninter = 0
            if nocc >= 2:
                ninter = int(lib.ref_des_uniq_strs(_p(inter), norb, C.c_int64(n), nocc, _p(strs)))
            inter = inter[:ninter].copy()
            ndl = max(1, (nvir + 2) * (nvir + 1) // 2)
            dd = np.zeros((max(ninter, 1), ndl, 4), dtype=np.int32)
            if ninter:
                lib.ref_des_des_linkstr_tril(_p(dd), norb, C.c_int64(n), nocc, _p(strs), C.c_int64(ninter), _p(inter), ndl)
            self.cd.append(cd); self.dd.append(dd); self.nlink.append(nlink); self.ndl.append(ndl); self.ninter.append(ninter)
        self.t_tables = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.hdiag = np.zeros(self.na * self.nb)
        lib.ref_make_hdiag(_p(self.hdiag), _p(self.h1), _p(self.eri), norb, C.c_int64(self.na), C.c_int64(self.nb),
                           _p(self.sa), _p(self.sb))
        self.t_hdiag = time.perf_counter() - t0
        self.h2e = np.zeros_like(self.eri)
        lib.ref_absorb_h1e(_p(self.h2e), _p(self.h1), _p(self.eri), norb, sum(self.nelec), C.c_double(0.5))

    def contract_2e(self, c):
        lib = load()
        c = np.ascontiguousarray(c, dtype=np.float64).ravel()
        out = np.zeros_like(c)
        lib.ref_contract_2e(
            _p(out), _p(c), _p(self.h2e), self.norb, self.nelec[0], self.nelec[1], C.c_int64(self.na), C.c_int64(self.nb),
            self.nlink[0], _p(self.cd[0]), self.nlink[1], _p(self.cd[1]),
            C.c_int64(self.ninter[0]), self.ndl[0], _p(self.dd[0]), C.c_int64(self.ninter[1]), self.ndl[1], _p(self.dd[1]),
        )
        return out

    def occupancies(self, c):
        lib = load()
        c = np.ascontiguousarray(c, dtype=np.float64).ravel()
        oa, ob = np.zeros(self.norb), np.zeros(self.norb)
        lib.ref_occupancies(_p(oa), _p(ob), _p(c), self.norb, C.c_int64(self.na), C.c_int64(self.nb), _p(self.sa), _p(self.sb))
        return oa, ob

    def dense_flops_per_sigma(self) -> float:
        """Flop count of the dense formulation actually executed (SURVEY 8d F_sigma)."""
        nn_s = self.norb * (self.norb + 1) // 2
        nn_a = self.norb * (self.norb - 1) // 2
        return 2.0 * self.na * self.nb * nn_s**2 + 2.0 * (self.ninter[0] * self.nb + self.ninter[1] * self.na) * nn_a**2


def solve_fermion_ref(ci_strs, hcore, eri, tol=1e-9, max_cycle=100, max_space=12):
    """Reference ``solve_fermion`` (fermion.py:745-845) without spin penalty: check strings, pyscf-flow
    Davidson on the restated contract_2e, energy as <c|H|c>, occupancies.  Returns (e, amps, occ, n_sigma)."""
    sa, sb = O.check_ci_strs(ci_strs)
    prob = RefProblem(hcore, eri, sa, sb)
    x0 = O.init_guess(prob.hdiag, prob.na, prob.nb, prob.nelec)
    conv, e, x, nsig = O.davidson_pyscf(prob.contract_2e, x0, prob.hdiag, tol=tol, max_cycle=max_cycle,
                                        max_space=max_space)
    x = x / np.linalg.norm(x)
    e_sci = float(x @ prob.contract_2e(x))
    return e_sci, x.reshape(prob.na, prob.nb), prob.occupancies(x), nsig + 1
