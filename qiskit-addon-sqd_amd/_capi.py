"""ctypes binding of ``libsqd_hip.so`` (C ABI in ``include/sqd_hip.h``).

This is the only place the package crosses into native code.  There is no CPU
fallback: if the HIP library is missing or fails to load, importing the solver
entry points raises ``SQDNativeError`` with build instructions.
"""

from __future__ import annotations

import ctypes as C
import os
import weakref
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_NAME = "libsqd_hip.so"
LIB_PATH = _HERE / "csrc" / LIB_NAME

EXPORTED_SYMBOLS = (
    "sqd_abi_version",
    "sqd_last_error",
    "sqd_device_count",
    "sqd_host_alloc",
    "sqd_host_free",
    "sqd_ctx_create",
    "sqd_ctx_destroy",
    "sqd_ctx_use_stream",
    "sqd_ctx_set_record_out",
    "sqd_ctx_set_enqueue_hook",
    "sqd_set_subspace",
    "sqd_set_subspace_rows",
    "sqd_sigma_rows_dev",
    "sqd_contract_ss_rows_dev",
    "sqd_hdiag_rows_dev",
    "sqd_ctx_sync",
    "sqd_shard_dav_begin",
    "sqd_shard_dav_pick",
    "sqd_shard_dav_sigma",
    "sqd_shard_dav_sigma_part",
    "sqd_shard_dav_dots",
    "sqd_shard_dav_residual",
    "sqd_shard_dav_orth",
    "sqd_shard_dav_iteration",
    "sqd_shard_dav_wait",
    "sqd_shard_dav_end",
    "sqd_solution_device_ptr",
    "sqd_solution_copy",
    "sqd_ctx_set_phase_timing",
    "sqd_ctx_set_async_state",
    "sqd_ctx_state_wait",
    "sqd_get_dims",
    "sqd_link_counts",
    "sqd_single_links",
    "sqd_double_links",
    "sqd_hdiag",
    "sqd_init_guess",
    "sqd_sigma",
    "sqd_contract_ss",
    "sqd_davidson_default_opts",
    "sqd_davidson",
    "sqd_observables",
    "sqd_solve",
    "sqd_solve_strings",
    "sqd_solve_batch",
    "sqd_batch_state",
    "sqd_batch_ctx",
    "sqd_energy",
    "sqd_spin_square",
    "sqd_rdm1s",
    "sqd_rdm2",
    "sqd_rdm2s",
    "sqd_time_sigma",
    "sqd_time_sigma_brackets",
    "sqd_time_dense",
    "sqd_sigma_bytes",
    "sqd_sigma_bytes_needed",
    "sqd_sigma_kernel",
    "sqd_pauli_count",
    "sqd_pauli_fill",
    "sqd_pauli_free",
    "sqd_recover_rows",
    "sqd_hamming_excess",
    "sqd_merge_rows",
    "sqd_unique_rows",
    "sqd_choice_replay",
    "sqd_hash_start",
    "sqd_hash_finish",
    "sqd_check_strings",
)


class SQDNativeError(RuntimeError):
    """Raised when libsqd_hip.so is unavailable or a native call fails."""


class DavidsonOpts(C.Structure):
    _fields_ = [
        ("tol", C.c_double),
        ("tol_residual", C.c_double),
        ("lindep", C.c_double),
        ("max_cycle", C.c_int),
        ("max_space", C.c_int),
        ("use_spin", C.c_int),
        ("ss", C.c_double),
        ("shift", C.c_double),
        ("verbose", C.c_int),
        ("time_sigma_every", C.c_int),
    ]


class DavidsonStats(C.Structure):
    _fields_ = [
        ("converged", C.c_int),
        ("iterations", C.c_int),
        ("n_sigma", C.c_int),
        ("e_davidson", C.c_double),
        ("residual", C.c_double),
        ("ms_total", C.c_double),
        ("ms_sigma", C.c_double),
        ("ms_setup", C.c_double),
        ("n_sigma_timed", C.c_int),
        ("ms_sigma_kernel", C.c_double),
        ("ms_event_overhead", C.c_double),
        ("n_eig_solves", C.c_int),
        ("n_eig_fallbacks", C.c_int),
        ("state_ticket", C.c_longlong),
    ]


_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u64p = C.POINTER(C.c_uint64)
_ctxp = C.c_void_p


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach prototypes for every symbol declared in include/sqd_hip.h."""
    lib.sqd_abi_version.restype = C.c_int
    lib.sqd_last_error.restype = C.c_char_p
    lib.sqd_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.sqd_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.sqd_host_free.argtypes = [C.c_void_p]
    lib.sqd_ctx_create.argtypes = [C.c_int, C.c_int, _dp, _dp, C.POINTER(_ctxp)]
    lib.sqd_ctx_destroy.argtypes = [_ctxp]
    lib.sqd_ctx_use_stream.argtypes = [_ctxp, C.c_void_p]
    lib.sqd_ctx_set_record_out.argtypes = [_ctxp, C.c_void_p, C.c_int64]
    lib.sqd_ctx_set_enqueue_hook.argtypes = [_ctxp, ENQUEUE_HOOK, C.c_void_p]
    lib.sqd_set_subspace.argtypes = [_ctxp, _u64p, C.c_int64, _u64p, C.c_int64]
    lib.sqd_set_subspace_rows.argtypes = [_ctxp, _u64p, C.c_int64, _u64p, C.c_int64, C.c_int64, C.c_int64]
    lib.sqd_sigma_rows_dev.argtypes = [_ctxp, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.sqd_contract_ss_rows_dev.argtypes = [_ctxp, C.c_void_p, C.c_void_p]
    lib.sqd_hdiag_rows_dev.argtypes = [_ctxp, C.c_void_p]
    lib.sqd_ctx_sync.argtypes = [_ctxp]
    lib.sqd_shard_dav_begin.argtypes = [_ctxp, C.POINTER(DavidsonOpts), C.POINTER(C.c_void_p)]
    lib.sqd_shard_dav_pick.argtypes = [_ctxp, C.POINTER(C.c_void_p)]
    lib.sqd_shard_dav_sigma.argtypes = [_ctxp, C.c_void_p]
    lib.sqd_shard_dav_sigma_part.argtypes = [_ctxp, C.c_void_p, C.c_int]
    lib.sqd_shard_dav_dots.argtypes = [_ctxp, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.sqd_shard_dav_residual.argtypes = [_ctxp, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.sqd_shard_dav_orth.argtypes = [_ctxp, C.POINTER(C.c_longlong)]
    lib.sqd_shard_dav_iteration.argtypes = [_ctxp, C.POINTER(C.c_longlong)]
    lib.sqd_shard_dav_wait.argtypes = [_ctxp, C.c_longlong, C.POINTER(C.c_int), _dp, _dp, C.POINTER(C.c_int)]
    lib.sqd_shard_dav_end.argtypes = [_ctxp, C.POINTER(C.c_void_p), C.POINTER(DavidsonStats)]
    lib.sqd_solution_device_ptr.argtypes = [_ctxp, C.POINTER(C.c_void_p)]
    lib.sqd_solution_copy.argtypes = [_ctxp, C.c_void_p]
    lib.sqd_ctx_set_phase_timing.argtypes = [_ctxp, C.c_int]
    lib.sqd_ctx_set_async_state.argtypes = [_ctxp, C.c_int]
    lib.sqd_ctx_state_wait.argtypes = [_ctxp, C.c_longlong]
    lib.sqd_get_dims.argtypes = [_ctxp, _i64p, _i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.sqd_link_counts.argtypes = [_ctxp, C.c_int, _i64p, _i64p]
    lib.sqd_single_links.argtypes = [_ctxp, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _dp]
    lib.sqd_double_links.argtypes = [_ctxp, C.c_int, _i32p, _i32p, _i32p, _i32p, _dp]
    lib.sqd_hdiag.argtypes = [_ctxp, _dp]
    lib.sqd_init_guess.argtypes = [_ctxp, _dp]
    lib.sqd_sigma.argtypes = [_ctxp, _dp, _dp, C.c_int, C.c_double, C.c_double]
    lib.sqd_contract_ss.argtypes = [_ctxp, _dp, _dp]
    lib.sqd_davidson_default_opts.argtypes = [C.POINTER(DavidsonOpts)]
    lib.sqd_davidson_default_opts.restype = None
    lib.sqd_davidson.argtypes = [_ctxp, C.POINTER(DavidsonOpts), _dp, _dp, C.POINTER(DavidsonStats)]
    lib.sqd_observables.argtypes = [_ctxp, _dp, _dp, _dp, _dp, _dp]
    lib.sqd_solve.argtypes = [_ctxp, C.POINTER(DavidsonOpts), _dp, _dp, C.POINTER(DavidsonStats), _dp, _dp, _dp, _dp]
    # (the hot entry point: array arguments are passed as plain addresses -- building a typed ctypes pointer per array
    # costs ~2 us each, five of them per solve)
    _vp = C.c_void_p
    lib.sqd_solve_strings.argtypes = [_ctxp, _vp, C.c_int64, _vp, C.c_int64, C.POINTER(DavidsonOpts), _vp, _vp,
                                      C.POINTER(DavidsonStats), _dp, _dp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.sqd_solve_batch.argtypes = [_ctxp, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(DavidsonOpts), _vp, _vp,
                                    C.POINTER(C.c_int), _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.sqd_batch_state.argtypes = [_ctxp, C.c_int, C.c_int, _vp]
    lib.sqd_batch_ctx.argtypes = [_ctxp, C.c_int, C.POINTER(_ctxp)]
    lib.sqd_energy.argtypes = [_ctxp, _dp, _dp]
    lib.sqd_spin_square.argtypes = [_ctxp, _dp, _dp]
    lib.sqd_rdm1s.argtypes = [_ctxp, _dp, _dp, _dp]
    lib.sqd_rdm2.argtypes = [_ctxp, _dp, _dp]
    lib.sqd_rdm2s.argtypes = [_ctxp, _dp, _dp, _dp, _dp]
    lib.sqd_time_sigma.argtypes = [_ctxp, C.c_int, C.c_int, C.c_double, C.c_double, _dp]
    lib.sqd_time_sigma_brackets.argtypes = [_ctxp, C.c_int, C.c_int, C.c_double, C.c_double, _dp]
    lib.sqd_time_dense.argtypes = [_ctxp, C.c_int, C.c_int, _dp, _dp]
    lib.sqd_sigma_bytes.argtypes = [_ctxp, _dp]
    lib.sqd_sigma_bytes_needed.argtypes = [_ctxp, _dp]
    lib.sqd_sigma_kernel.argtypes = [_ctxp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.sqd_pauli_count.argtypes = [C.c_int, _u64p, C.c_int64, C.c_int, _u64p, _i64p, _u64p, _dp, _i64p, _i64p,
                                    C.POINTER(_ctxp)]
    lib.sqd_pauli_fill.argtypes = [_ctxp, _i64p, _dp, _dp]
    lib.sqd_pauli_free.argtypes = [_ctxp]
    lib.sqd_recover_rows.argtypes = [C.POINTER(C.c_uint8), C.c_int64, C.c_int, _i64p, C.c_int64, _dp, _dp, _dp, _dp,
                                     C.c_int, C.c_int, _dp, C.c_int64, _i64p]
    lib.sqd_hamming_excess.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, _i64p]
    lib.sqd_unique_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, _i64p]
    lib.sqd_merge_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _i64p, C.c_int]
    lib.sqd_choice_replay.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, _i64p]
    lib.sqd_hash_start.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.sqd_hash_finish.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.sqd_check_strings.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("sqd_last_error", "sqd_davidson_default_opts"):
            fn.restype = C.c_int
    return lib


_LIB: C.CDLL | None = None


ENQUEUE_HOOK = C.CFUNCTYPE(None, C.c_void_p)  # sqd_enqueue_hook


def load_library() -> C.CDLL:
    """Load the hipcc-built library that lives in-tree.  No fallback of any kind."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not LIB_PATH.exists():
        raise SQDNativeError(
            f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). qiskit_addon_sqd_amd has no CPU fallback."
        )
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as /opt/rocm's).  A process that ends up with both
    # loses the device in whichever initialises second ("no ROCm-capable device is detected"), so torch --
    # this package's plumbing for streams and torch.distributed -- is imported first when it is installed:
    # the library then binds to the runtime torch has already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch-free deployments use the system runtime alone
        pass
    try:
        lib = C.CDLL(str(LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)
    except OSError as exc:  # pragma: no cover - depends on the box
        raise SQDNativeError(f"failed to load {LIB_PATH}: {exc}") from exc
    _LIB = bind(lib)
    return _LIB


class _PinnedPool:
    """Result arrays backed by page-locked host memory (``sqd_host_alloc``), recycled when the numpy array that wraps a
    block is garbage collected.  The DMA engine writes the amplitudes of a solve straight into such an array; an
    ordinary ``np.empty`` of 0.8 MB costs a staging copy plus a fresh mmap with ~200 first-touch page faults per solve."""

    KEEP_BYTES = 256 << 20  # page-locked bytes kept for reuse (not swappable, shared by the node: bounded)
    CLASS_MAX = 64 << 20    # larger blocks are allocated at their exact size and returned to the OS when released

    def __init__(self):
        import threading

        # re-entrant: _release runs from a weakref finaliser, which the garbage collector may start on a thread that
        # is inside empty() / _release() already
        self._lock = threading.RLock()
        self._free: dict[int, list[int]] = {}
        self._kept = 0

    def empty(self, shape) -> np.ndarray:
        n = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            n *= int(d)
        nbytes = max(8 * n, 8)
        # power-of-two size classes up to CLASS_MAX, the exact size (to the page) beyond
        size = 1 << max(12, (nbytes - 1).bit_length()) if nbytes <= self.CLASS_MAX else (nbytes + 4095) & ~4095
        ptr = None
        with self._lock:
            blocks = self._free.get(size)
            if blocks:
                ptr = blocks.pop()
                self._kept -= size
        if ptr is None:
            out = C.c_void_p()
            rc = load_library().sqd_host_alloc(size, C.byref(out))
            if rc != 0 or not out.value:
                return np.empty(shape)  # page-locked memory exhausted: a plain array still works (staged copy)
            ptr = out.value
        buf = (C.c_double * n).from_address(ptr)
        fin = weakref.finalize(buf, self._release, ptr, size)
        fin.atexit = False  # at interpreter exit the process' memory goes away anyway; no HIP calls during teardown
        return np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)

    def _release(self, ptr, size):
        with self._lock:
            if size <= self.CLASS_MAX and self._kept + size <= self.KEEP_BYTES:
                blocks = self._free.get(size)
                if blocks is None:
                    blocks = self._free[size] = []
                blocks.append(ptr)
                self._kept += size
                return
        lib = _LIB
        if lib is not None:
            lib.sqd_host_free(C.c_void_p(ptr))


_PINNED = _PinnedPool()


def pinned_empty(shape) -> np.ndarray:
    """Uninitialised float64 array in page-locked host memory (falls back to ``np.empty``)."""
    return _PINNED.empty(shape)


def _as_f64(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and out.shape != shape:
        raise ValueError(f"expected array of shape {shape}, got {out.shape}")
    return out


def _ptr(a: np.ndarray, typ=_dp):
    return a.ctypes.data_as(typ)


def _addr(a: np.ndarray) -> int:
    """Address of a (contiguous) array's first element, for arguments declared ``c_void_p``."""
    return a.__array_interface__["data"][0]


def strings_to_u64(strs, validated: bool = False) -> np.ndarray:
    """CI strings (int64 / uint64 / python ints) -> contiguous uint64 array (bit pattern preserved).  ``validated``: the
    caller has checked the list (ascending, non-negative: ``fermion._check_ci_strs``) -- no pass over it here."""
    arr = np.asarray(strs)
    if validated and arr.dtype in (np.int64, np.uint64) and arr.flags.c_contiguous:
        return arr if arr.dtype == np.uint64 else arr.view(np.uint64)
    if arr.dtype == np.int64 and arr.flags.c_contiguous and (arr.size == 0 or (arr[0] >= 0 and arr[-1] >= 0 and arr.min() >= 0)):
        return arr.view(np.uint64)  # same bits, no copy
    if arr.dtype == object:
        arr = np.array([int(x) for x in arr], dtype=np.uint64)
    elif arr.dtype != np.uint64:
        if arr.size and np.issubdtype(arr.dtype, np.signedinteger) and (arr < 0).any():
            raise ValueError("CI strings must be non-negative integers")
        arr = arr.astype(np.uint64)
    return np.ascontiguousarray(arr)


def record_width(norb: int) -> int:
    """Doubles in the raw observables record of ``Context.set_record_out``."""
    return 5 + 2 * int(norb)


def results_from_record(rec: np.ndarray, norb: int, spin_sq: float | None, shift: float, nelec) -> tuple:
    """(energy, occ_a, occ_b) from a raw observables record, with the arithmetic -- operation for operation -- of the
    native ``solve_collect`` (sqd_capi.hip), so that a record that travelled through a collective gives the bits the
    solving rank got from its own call."""
    e_dav, cs2c, cc, tt_raw = float(rec[0]), float(rec[2]), float(rec[3]), float(rec[4 + 2 * norb])
    form = 0
    if spin_sq is not None:
        szh = 0.5 * abs(int(nelec[0]) - int(nelec[1]))
        form = 1 if float(spin_sq) < szh * (szh + 1.0) + 0.1 else 2
    ct, tt = cs2c / cc, tt_raw / cc
    penalty = 0.0
    if form == 1:
        penalty = ct - float(spin_sq)
    elif form == 2:
        ss = float(spin_sq)
        penalty = tt - 2.0 * ss * ct + ss * ss
    energy = e_dav - (float(shift) * penalty if form else 0.0)
    return energy, rec[4 : 4 + norb] / cc, rec[4 + norb : 4 + 2 * norb] / cc


class Context:
    """RAII wrapper of one ``sqd_ctx`` (one device, one stream, one Hamiltonian)."""

    def __init__(self, hcore, eri, device: int = 0, lib: C.CDLL | None = None):
        self._lib = lib if lib is not None else load_library()
        self._h = _ctxp()
        hcore = _as_f64(hcore)
        if hcore.ndim != 2 or hcore.shape[0] != hcore.shape[1]:
            raise ValueError("hcore must be a square matrix")
        self.norb = int(hcore.shape[0])
        eri = _as_f64(eri)
        if eri.size != self.norb**4:
            raise ValueError(f"eri must have norb**4 = {self.norb**4} elements (chemist order, no symmetry packing)")
        self.device = int(device)
        self.na = self.nb = 0
        self.nelec = (0, 0)
        self._check(self._lib.sqd_ctx_create(self.device, self.norb, _ptr(hcore), _ptr(eri), C.byref(self._h)))

    # -- plumbing
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.sqd_last_error()
            msg = msg.decode() if msg else "unknown error"
            if rc == -1:
                raise ValueError(msg)
            raise SQDNativeError(f"libsqd_hip error {rc}: {msg}")

    def set_enqueue_hook(self, fn):
        """``fn()`` is called once by every following ``solve`` / ``solve_batch`` of this context when its last kernel
        has been enqueued and before it waits (``sqd_ctx_set_enqueue_hook``): the place to enqueue a collective on the
        shared stream.  ``None`` removes the hook.  Exceptions raised by ``fn`` are re-raised by ``raise_hook_error``."""
        if fn is None:
            self._hook_fn = None
            self._check(self._lib.sqd_ctx_set_enqueue_hook(self._h, C.cast(None, ENQUEUE_HOOK), None))
            return
        if getattr(self, "_hook_c", None) is None:
            # ONE native-callable thunk per context, made once: a new ctypes closure per call is a libffi allocation
            # plus cyclic garbage, and the collector's full passes showed up as 30 ms steps in the N > 1 bench
            self._hook_error = None
            self._hook_c = ENQUEUE_HOOK(self._hook_trampoline)
        self._hook_fn = fn
        self._check(self._lib.sqd_ctx_set_enqueue_hook(self._h, self._hook_c, None))

    def _hook_trampoline(self, _user):
        fn = self._hook_fn
        if fn is None:
            return
        try:
            fn()
        except BaseException as exc:  # (must not propagate through the C frames)
            self._hook_error = exc

    def raise_hook_error(self):
        exc, self._hook_error = getattr(self, "_hook_error", None), None
        if exc is not None:
            raise exc

    def use_stream(self, stream_handle: int):
        """Enqueue all further work of this context on a caller-owned HIP stream (``hipStream_t`` as an integer,
        e.g. ``torch.cuda.Stream().cuda_stream``); the stream must outlive the context."""
        self._check(self._lib.sqd_ctx_use_stream(self._h, C.c_void_p(int(stream_handle))))

    def set_record_out(self, device_ptr: int | None, stride: int = 0):
        """Device address (``tensor.data_ptr()``) where the following solves also leave their raw observables record
        ``{e_davidson, c.Hc, c.S2c, c.c, occ_a, occ_b, |S2 c|^2}`` (``record_width(norb)`` doubles; batch p of
        ``solve_batch`` at ``device_ptr + 8 * p * stride``); ``None`` switches it off."""
        self._check(self._lib.sqd_ctx_set_record_out(self._h, C.c_void_p(int(device_ptr) if device_ptr else None), int(stride)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.sqd_ctx_destroy(self._h)
            self._h = _ctxp()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- subspace
    def set_subspace(self, strs_a, strs_b):
        a = strings_to_u64(strs_a)
        b = strings_to_u64(strs_b)
        self._check(self._lib.sqd_set_subspace(self._h, _ptr(a, _u64p), a.size, _ptr(b, _u64p), b.size))
        na, nb = C.c_int64(), C.c_int64()
        ea, eb = C.c_int(), C.c_int()
        self._check(self._lib.sqd_get_dims(self._h, C.byref(na), C.byref(nb), C.byref(ea), C.byref(eb)))
        self.na, self.nb = int(na.value), int(nb.value)
        self.nelec = (int(ea.value), int(eb.value))
        self.rows = (0, self.na)

    def set_subspace_rows(self, strs_a, strs_b, row0: int, row1: int):
        """This context serves alpha rows [row0, row1) of the subspace (intra-solve sharding)."""
        a = strings_to_u64(strs_a)
        b = strings_to_u64(strs_b)
        self._check(self._lib.sqd_set_subspace_rows(self._h, _ptr(a, _u64p), a.size, _ptr(b, _u64p), b.size,
                                                    int(row0), int(row1)))
        na, nb = C.c_int64(), C.c_int64()
        ea, eb = C.c_int(), C.c_int()
        self._check(self._lib.sqd_get_dims(self._h, C.byref(na), C.byref(nb), C.byref(ea), C.byref(eb)))
        self.na, self.nb = int(na.value), int(nb.value)
        self.nelec = (int(ea.value), int(eb.value))
        self.rows = (int(row0), int(row1))

    def sigma_rows_dev(self, c_full_ptr: int, out_rows_ptr: int, use_spin: int = 0, ss: float = 0.0, shift: float = 0.0):
        """Enqueue sigma rows [row0, row1) <- full vector; both arguments are DEVICE addresses (``tensor.data_ptr()``)."""
        self._check(self._lib.sqd_sigma_rows_dev(self._h, C.c_void_p(int(c_full_ptr)), C.c_void_p(int(out_rows_ptr)),
                                                 int(use_spin), float(ss), float(shift)))

    def contract_ss_rows_dev(self, c_full_ptr: int, out_rows_ptr: int):
        self._check(self._lib.sqd_contract_ss_rows_dev(self._h, C.c_void_p(int(c_full_ptr)), C.c_void_p(int(out_rows_ptr))))

    def hdiag_rows_dev(self, out_rows_ptr: int):
        self._check(self._lib.sqd_hdiag_rows_dev(self._h, C.c_void_p(int(out_rows_ptr))))

    def fetch_solution(self) -> np.ndarray:
        """The resident Davidson solution of the latest solve on this context, copied to a page-locked host array."""
        import ctypes as _C

        out = pinned_empty((self.na, self.nb))
        load_library()
        # (hipMemcpy through the library's own runtime: sqd_batch_state serves sub-contexts, this the context itself)
        self._check(self._lib.sqd_solution_copy(self._h, _addr(out)))
        return out

    # -- row-sharded Davidson, stage by stage (device addresses in and out; the caller owns the collectives)
    def shard_dav_begin(self, *, tol=1e-9, tol_residual=None, lindep=1e-14, max_cycle=100, max_space=12, spin_sq=None,
                        shift=0.2) -> int:
        opts = DavidsonOpts()
        self._lib.sqd_davidson_default_opts(C.byref(opts))
        opts.tol, opts.lindep, opts.max_cycle, opts.max_space = tol, lindep, int(max_cycle), int(max_space)
        opts.tol_residual = float(tol_residual) if tol_residual else 0.0
        if spin_sq is not None:
            opts.use_spin, opts.ss, opts.shift = 3, float(spin_sq), float(shift)
        out = C.c_void_p()
        self._check(self._lib.sqd_shard_dav_begin(self._h, C.byref(opts), C.byref(out)))
        return int(out.value)

    def shard_dav_pick(self) -> int:
        out = C.c_void_p()
        self._check(self._lib.sqd_shard_dav_pick(self._h, C.byref(out)))
        return int(out.value)

    def shard_dav_sigma(self, full_ptr: int):
        self._check(self._lib.sqd_shard_dav_sigma(self._h, C.c_void_p(int(full_ptr))))

    def shard_dav_dots(self):
        out, n = C.c_void_p(), C.c_int()
        self._check(self._lib.sqd_shard_dav_dots(self._h, C.byref(out), C.byref(n)))
        return int(out.value), int(n.value)

    def shard_dav_residual(self):
        out, n = C.c_void_p(), C.c_int()
        self._check(self._lib.sqd_shard_dav_residual(self._h, C.byref(out), C.byref(n)))
        return int(out.value), int(n.value)

    def shard_dav_orth(self) -> int:
        """Enqueue the last stage of an iteration; returns the ticket ``shard_dav_wait`` takes."""
        t = C.c_longlong()
        self._check(self._lib.sqd_shard_dav_orth(self._h, C.byref(t)))
        return int(t.value)

    def shard_dav_wait(self, ticket: int):
        stop, m = C.c_int(), C.c_int()
        e, rr = C.c_double(), C.c_double()
        self._check(self._lib.sqd_shard_dav_wait(self._h, int(ticket), C.byref(stop), C.byref(e), C.byref(rr), C.byref(m)))
        return bool(stop.value), float(e.value), float(rr.value), int(m.value)

    def shard_dav_stages(self):
        """The five per-iteration stage calls with their ctypes arguments built once (``pick() -> ptr``, ``sigma(ptr)``,
        ``dots() -> (ptr, n)``, ``residual() -> (ptr, n)``, ``orth() -> ticket``): an iteration of the row-sharded solver
        is ~85 us of kernels at batch size, and the generic wrappers above cost the host more than that per iteration."""
        lib, h, check = self._lib, self._h, self._check
        out, n, t = C.c_void_p(), C.c_int(), C.c_longlong()
        r_out, r_n, r_t = C.byref(out), C.byref(n), C.byref(t)
        f_pick, f_sigma, f_dots, f_res, f_orth = (lib.sqd_shard_dav_pick, lib.sqd_shard_dav_sigma, lib.sqd_shard_dav_dots,
                                                  lib.sqd_shard_dav_residual, lib.sqd_shard_dav_orth)

        def pick():
            rc = f_pick(h, r_out)
            if rc:
                check(rc)
            return out.value

        def sigma(ptr, part=0):
            rc = f_sigma(h, ptr) if part == 0 else lib.sqd_shard_dav_sigma_part(h, ptr, part)
            if rc:
                check(rc)

        def dots():
            rc = f_dots(h, r_out, r_n)
            if rc:
                check(rc)
            return out.value, n.value

        def residual():
            rc = f_res(h, r_out, r_n)
            if rc:
                check(rc)
            return out.value, n.value

        def orth():
            rc = f_orth(h, r_t)
            if rc:
                check(rc)
            return t.value

        return pick, sigma, dots, residual, orth

    def shard_dav_iteration_call(self):
        """``iteration() -> ticket``: the five stages as ONE native call (a group of one rank, where no collective sits
        between them), arguments built once."""
        lib, h, check = self._lib, self._h, self._check
        t = C.c_longlong()
        r_t = C.byref(t)
        f = lib.sqd_shard_dav_iteration

        def iteration():
            rc = f(h, r_t)
            if rc:
                check(rc)
            return t.value

        return iteration

    def shard_dav_end(self):
        out = C.c_void_p()
        stats = DavidsonStats()
        self._check(self._lib.sqd_shard_dav_end(self._h, C.byref(out), C.byref(stats)))
        return int(out.value), {f[0]: getattr(stats, f[0]) for f in DavidsonStats._fields_}

    def solution_device_ptr(self) -> int:
        """Device address of the resident Davidson solution (valid until the next solve on this context)."""
        out = C.c_void_p()
        self._check(self._lib.sqd_solution_device_ptr(self._h, C.byref(out)))
        return int(out.value)

    def set_async_state(self, on: bool):
        """``solve`` returns when the results are on the host; the amplitudes follow into the (page-locked) result
        buffer -- ``stats["state_ticket"]`` > 0 then, and ``state_wait(ticket)`` says when they are complete."""
        if getattr(self, "_async_state", False) != bool(on):
            self._check(self._lib.sqd_ctx_set_async_state(self._h, 1 if on else 0))
            self._async_state = bool(on)

    def state_wait(self, ticket: int):
        if getattr(self, "_h", None) is not None and self._h.value:  # (a closed context has drained its stream)
            self._check(self._lib.sqd_ctx_state_wait(self._h, int(ticket)))

    def set_phase_timing(self, on: bool):
        """Fill ``ms_setup`` / ``ms_total`` of the Davidson statistics (HIP events around the table build and the
        Davidson run of every following solve; costs stream bubbles, off by default)."""
        self._check(self._lib.sqd_ctx_set_phase_timing(self._h, 1 if on else 0))

    def sync(self):
        self._check(self._lib.sqd_ctx_sync(self._h))

    def link_counts(self, spin: int):
        ns, nd = C.c_int64(), C.c_int64()
        self._check(self._lib.sqd_link_counts(self._h, spin, C.byref(ns), C.byref(nd)))
        return int(ns.value), int(nd.value)

    def single_links(self, spin: int) -> dict:
        ns, _ = self.link_counts(spin)
        out = {k: np.zeros(ns, dtype=np.int32) for k in ("tgt", "src", "cre", "des", "pair", "sign")}
        val = np.zeros(ns)
        self._check(
            self._lib.sqd_single_links(
                self._h, spin, *[_ptr(out[k], _i32p) for k in ("tgt", "src", "cre", "des", "pair", "sign")], _ptr(val)
            )
        )
        out["value"] = val
        return out

    def double_links(self, spin: int) -> dict:
        _, nd = self.link_counts(spin)
        tgt = np.zeros(nd, dtype=np.int32)
        src = np.zeros(nd, dtype=np.int32)
        orbs = np.zeros((nd, 4), dtype=np.int32)
        sign = np.zeros(nd, dtype=np.int32)
        val = np.zeros(nd)
        self._check(
            self._lib.sqd_double_links(
                self._h, spin, _ptr(tgt, _i32p), _ptr(src, _i32p), _ptr(orbs, _i32p), _ptr(sign, _i32p), _ptr(val)
            )
        )
        return dict(tgt=tgt, src=src, orbs=orbs, sign=sign, value=val)

    def hdiag(self) -> np.ndarray:
        rows = getattr(self, "rows", (0, self.na))
        out = np.empty((rows[1] - rows[0], self.nb))
        self._check(self._lib.sqd_hdiag(self._h, _ptr(out)))
        return out

    def init_guess(self) -> np.ndarray:
        """The Davidson start vector used when no ``ci0`` is given (pyscf ``get_init_guess``), normalised."""
        out = np.empty((self.na, self.nb))
        self._check(self._lib.sqd_init_guess(self._h, _ptr(out)))
        return out

    # -- operators
    def sigma(self, c, use_spin: int = 0, ss: float = 0.0, shift: float = 0.0) -> np.ndarray:
        c = _as_f64(c).reshape(self.na, self.nb)
        out = np.empty_like(c)
        self._check(self._lib.sqd_sigma(self._h, _ptr(c), _ptr(out), use_spin, ss, shift))
        return out

    def contract_ss(self, c) -> np.ndarray:
        c = _as_f64(c).reshape(self.na, self.nb)
        out = np.empty_like(c)
        self._check(self._lib.sqd_contract_ss(self._h, _ptr(c), _ptr(out)))
        return out

    def davidson(
        self,
        ci0=None,
        *,
        tol: float = 1e-9,
        tol_residual: float | None = None,
        lindep: float = 1e-14,
        max_cycle: int = 100,
        max_space: int = 12,
        spin_sq: float | None = None,
        shift: float = 0.2,
        verbose: int = 0,
        fetch: bool = True,
        time_sigma_every: int = 0,
        observables: bool = False,
        spin_square: bool = True,
    ):
        """Ground state of the projected Hamiltonian.  Returns (amps, stats); with ``observables=True`` the
        fused native call ``sqd_solve`` is used and (amps, stats, (energy, spin_square, occ_a, occ_b)) is
        returned -- the observables kernel also writes the amplitudes into the (page-locked) result buffer."""
        opts = DavidsonOpts()
        self._lib.sqd_davidson_default_opts(C.byref(opts))
        opts.tol, opts.lindep, opts.max_cycle, opts.max_space = tol, lindep, int(max_cycle), int(max_space)
        opts.verbose = int(verbose)
        opts.time_sigma_every = int(time_sigma_every)
        opts.tol_residual = float(tol_residual) if tol_residual else 0.0
        if spin_sq is not None:
            opts.use_spin, opts.ss, opts.shift = 3, float(spin_sq), float(shift)
        stats = DavidsonStats()
        amps = pinned_empty((self.na, self.nb)) if fetch else None
        ci0p = None
        if ci0 is not None:
            ci0 = _as_f64(ci0).reshape(self.na, self.nb)
            ci0p = _ptr(ci0)
        if observables:
            # spin_square=False: <S^2> is not asked for (the sci_solver seam never reads it); without a spin
            # penalty no sigma build at all then follows the Davidson
            e, s2 = C.c_double(), C.c_double()
            occ_a, occ_b = np.empty(self.norb), np.empty(self.norb)
            self._check(
                self._lib.sqd_solve(self._h, C.byref(opts), ci0p, _ptr(amps) if fetch else None, C.byref(stats),
                                    C.byref(e), C.byref(s2) if spin_square else None, _ptr(occ_a), _ptr(occ_b))
            )
            return (amps, {f[0]: getattr(stats, f[0]) for f in DavidsonStats._fields_},
                    (e.value, s2.value if spin_square else None, occ_a, occ_b))
        self._check(
            self._lib.sqd_davidson(self._h, C.byref(opts), ci0p, _ptr(amps) if fetch else None, C.byref(stats))
        )
        return amps, {f[0]: getattr(stats, f[0]) for f in DavidsonStats._fields_}

    def solve(self, strs_a, strs_b, ci0=None, *, tol: float = 1e-9, tol_residual: float | None = None,
              lindep: float = 1e-14, max_cycle: int = 100, max_space: int = 12, spin_sq: float | None = None,
              shift: float = 0.2, verbose: int = 0, time_sigma_every: int = 0, spin_square: bool = True,
              pageable_result: bool = False, fetch: bool = True, validated: bool = False):
        """``set_subspace`` + ``davidson(observables=True)`` in one native call (``sqd_solve_strings``).
        Returns (amps, stats, (energy, spin_square | None, occ_a, occ_b))."""
        a = strings_to_u64(strs_a, validated)
        b = strings_to_u64(strs_b, validated)
        tmpl = self.__dict__.get("_opts_template")
        if tmpl is None:  # the library's defaults, asked for once per context
            d0 = DavidsonOpts()
            self._lib.sqd_davidson_default_opts(C.byref(d0))
            tmpl = self._opts_template = bytes(d0)
        opts = DavidsonOpts.from_buffer_copy(tmpl)
        opts.tol, opts.lindep, opts.max_cycle, opts.max_space = tol, lindep, int(max_cycle), int(max_space)
        opts.verbose = int(verbose)
        opts.time_sigma_every = int(time_sigma_every)
        opts.tol_residual = float(tol_residual) if tol_residual else 0.0
        if spin_sq is not None:
            opts.use_spin, opts.ss, opts.shift = 3, float(spin_sq), float(shift)
        stats = DavidsonStats()
        # (pageable_result: an ordinary numpy buffer for the state, the way a C caller without sqd_host_alloc would
        # pass it -- the copy-stream path of sqd_solve instead of the kernel-written one; tests compare the two)
        # (fetch=False: the state stays on the device -- solution_device_ptr() / a later davidson-free copy)
        amps = None if not fetch else (np.empty((a.size, b.size)) if pageable_result else pinned_empty((a.size, b.size)))
        ci0p = None
        if ci0 is not None:
            ci0 = _as_f64(ci0).reshape(a.size, b.size)
            ci0p = _addr(ci0)
        e, s2 = C.c_double(), C.c_double()
        ea, eb = C.c_int(), C.c_int()
        occ = np.empty((2, self.norb))
        occ_a, occ_b = occ[0], occ[1]
        base = _addr(occ)
        self._check(
            self._lib.sqd_solve_strings(self._h, _addr(a), a.size, _addr(b), b.size, C.byref(opts), ci0p,
                                        _addr(amps) if amps is not None else None, C.byref(stats), C.byref(e), C.byref(s2) if spin_square else None,
                                        base, base + 8 * self.norb, C.byref(ea), C.byref(eb))
        )
        self.na, self.nb = int(a.size), int(b.size)
        self.nelec = (int(ea.value), int(eb.value))
        self.rows = (0, self.na)
        return (amps, {f[0]: getattr(stats, f[0]) for f in DavidsonStats._fields_},
                (e.value, s2.value if spin_square else None, occ_a, occ_b))

    def solve_batch(self, ci_strings, *, tol: float = 1e-9, tol_residual: float | None = None, lindep: float = 1e-14,
                    max_cycle: int = 100, max_space: int = 12, spin_sq: float | None = None, shift: float = 0.2,
                    spin_square: bool = False, fetch: str = "best"):
        """The whole list of subspaces as ONE batched solve (``sqd_solve_batch``): tables, every Davidson round and the
        observables of all of them advance in the same kernel launches.  ``fetch``: ``"best"`` -- only the state of the
        lowest-energy batch comes to the host (the others stay on the device until the next batched solve;
        ``batch_state(i)`` copies one out), ``"all"``, or ``"none"``.  Returns a dict with ``energy[n]``,
        ``spin_square[n] | None``, ``occ_a[n, norb]``, ``occ_b[n, norb]``, ``stats[n]``, ``nelec[n]``, ``best`` and
        ``amps`` (list: arrays, or None where the state was left on the device)."""
        n = len(ci_strings)
        if n < 1:
            raise ValueError("empty batch list")
        a_arr = [strings_to_u64(a) for a, _ in ci_strings]
        b_arr = [strings_to_u64(b) for _, b in ci_strings]
        pa = (C.c_void_p * n)(*[_addr(x) for x in a_arr])
        pb = (C.c_void_p * n)(*[_addr(x) for x in b_arr])
        na = (C.c_int64 * n)(*[x.size for x in a_arr])
        nb = (C.c_int64 * n)(*[x.size for x in b_arr])
        opts = DavidsonOpts()
        self._lib.sqd_davidson_default_opts(C.byref(opts))
        opts.tol, opts.lindep, opts.max_cycle, opts.max_space = tol, lindep, int(max_cycle), int(max_space)
        opts.tol_residual = float(tol_residual) if tol_residual else 0.0
        if spin_sq is not None:
            opts.use_spin, opts.ss, opts.shift = 3, float(spin_sq), float(shift)
        stats = (DavidsonStats * n)()
        e = np.empty(n)
        s2 = np.empty(n) if spin_square else None
        occ = np.empty((2, n, self.norb))
        ea, eb = (C.c_int * n)(), (C.c_int * n)()
        amps: list = [None] * n
        pamps = None
        best_buf, best = None, C.c_int(-1)
        if fetch == "all":
            amps = [pinned_empty((a_arr[i].size, b_arr[i].size)) for i in range(n)]
            pamps = (C.c_void_p * n)(*[_addr(x) for x in amps])
        elif fetch == "best":
            best_buf = pinned_empty(max(a_arr[i].size * b_arr[i].size for i in range(n)))
        elif fetch != "none":
            raise ValueError("fetch must be 'best', 'all' or 'none'")
        # (the generation moves only with a call that succeeds: a failing call leaves the resident solutions of the
        # previous calls where they were -- sqd_solve_batch rotates them back)
        self._check(
            self._lib.sqd_solve_batch(self._h, n, C.addressof(pa), C.addressof(na), C.addressof(pb), C.addressof(nb),
                                      C.byref(opts), C.addressof(pamps) if pamps is not None else None,
                                      _addr(best_buf) if best_buf is not None else None, C.byref(best),
                                      C.addressof(stats), _addr(e), _addr(s2) if s2 is not None else None, _addr(occ[0]),
                                      _addr(occ[1]), C.addressof(ea), C.addressof(eb))
        )
        self._batch_gen = getattr(self, "_batch_gen", 0) + 1
        self._batch_shapes_prev = getattr(self, "_batch_shapes", [])
        self._batch_shapes = [(a_arr[i].size, b_arr[i].size) for i in range(n)]
        if best_buf is not None:
            w = int(best.value)
            sh = self._batch_shapes[w]
            amps[w] = best_buf[: sh[0] * sh[1]].reshape(sh)
        names = [f[0] for f in DavidsonStats._fields_]
        return {
            "energy": e, "spin_square": s2, "occ_a": occ[0], "occ_b": occ[1],
            "stats": [{k: getattr(stats[i], k) for k in names} for i in range(n)],
            "nelec": [(int(ea[i]), int(eb[i])) for i in range(n)], "best": int(best.value), "amps": amps,
            "generation": self._batch_gen,
        }

    def batch_sub(self, index: int) -> "Context":
        """Borrowed handle on the sub-context that holds batch ``index`` of the latest ``solve_batch`` (tables + resident
        solution): ``link_counts``, ``sigma_kernel``, ``sigma_bytes``, RDMs ... until the next batched solve."""
        sub = object.__new__(Context)
        sub._lib, sub._h, sub.norb, sub.device = self._lib, _ctxp(), self.norb, self.device
        self._check(self._lib.sqd_batch_ctx(self._h, int(index), C.byref(sub._h)))
        sub.na, sub.nb = self._batch_shapes[index]
        sub.rows = (0, sub.na)
        sub.nelec = (0, 0)
        sub._owner = self       # keeps the parent alive
        sub.close = lambda: None  # owned by the parent context
        return sub

    def batch_state(self, index: int, generation: int | None = None) -> np.ndarray:
        """Amplitudes of batch ``index`` of a ``solve_batch`` from their device-resident copy: of the latest call, or
        (``generation`` = that call's ``"generation"``) of the one before it -- a slot keeps two solutions."""
        age = 0 if generation is None else self._batch_gen - int(generation)
        if age not in (0, 1):
            raise SQDNativeError("that batched solve's states are no longer resident on the device")
        sh = (self._batch_shapes if age == 0 else self._batch_shapes_prev)[index]
        out = pinned_empty(sh)
        self._check(self._lib.sqd_batch_state(self._h, int(index), age, _addr(out)))
        return out

    # -- observables (amps=None -> resident Davidson solution)
    def _state(self, amps):
        if amps is None:
            return None, None
        a = _as_f64(amps).reshape(self.na, self.nb)
        return a, _ptr(a)

    def observables(self, amps=None):
        """(energy, spin_square, occ_a, occ_b) of the state in one native call / one device sync."""
        keep, p = self._state(amps)
        e, s2 = C.c_double(), C.c_double()
        oa, ob = np.empty(self.norb), np.empty(self.norb)
        self._check(self._lib.sqd_observables(self._h, p, C.byref(e), C.byref(s2), _ptr(oa), _ptr(ob)))
        return float(e.value), float(s2.value), oa, ob

    def energy(self, amps=None) -> float:
        keep, p = self._state(amps)
        out = C.c_double()
        self._check(self._lib.sqd_energy(self._h, p, C.byref(out)))
        return float(out.value)

    def spin_square(self, amps=None) -> float:
        keep, p = self._state(amps)
        out = C.c_double()
        self._check(self._lib.sqd_spin_square(self._h, p, C.byref(out)))
        return float(out.value)

    def rdm1s(self, amps=None):
        keep, p = self._state(amps)
        a = np.empty((self.norb, self.norb))
        b = np.empty((self.norb, self.norb))
        self._check(self._lib.sqd_rdm1s(self._h, p, _ptr(a), _ptr(b)))
        return a, b

    def rdm2(self, amps=None) -> np.ndarray:
        keep, p = self._state(amps)
        out = np.empty((self.norb,) * 4)
        self._check(self._lib.sqd_rdm2(self._h, p, _ptr(out)))
        return out

    def rdm2s(self, amps=None):
        """(dm2aa, dm2ab, dm2bb), pyscf ``make_rdm2s`` convention."""
        keep, p = self._state(amps)
        out = [np.empty((self.norb,) * 4) for _ in range(3)]
        self._check(self._lib.sqd_rdm2s(self._h, p, *[_ptr(o) for o in out]))
        return tuple(out)

    # -- benchmark hooks
    def time_sigma(self, reps: int = 10, use_spin: int = 0, ss: float = 0.0, shift: float = 0.0) -> float:
        out = C.c_double()
        self._check(self._lib.sqd_time_sigma(self._h, reps, use_spin, ss, shift, C.byref(out)))
        return float(out.value)

    def time_sigma_brackets(self, reps: int = 200, use_spin: int = 0, ss: float = 0.0, shift: float = 0.0) -> dict:
        """HIP-event bracket around every one of ``reps`` launches of the dominant sigma kernel, an empty bracket behind each:
        mean / median of both in ms (the roofline leg of bench.py, run behind the timed region)."""
        out = (C.c_double * 4)()
        self._check(self._lib.sqd_time_sigma_brackets(self._h, reps, use_spin, ss, shift, out))
        return {"kernel_mean_ms": out[0], "kernel_median_ms": out[1], "empty_mean_ms": out[2], "empty_median_ms": out[3], "launches": reps}

    def time_dense(self, reps: int = 10, copies: int = 1):
        """(ms per launch, flops per launch) of the matrix-core same-spin product alone, ``copies`` problems per launch."""
        ms, fl = C.c_double(), C.c_double()
        self._check(self._lib.sqd_time_dense(self._h, reps, copies, C.byref(ms), C.byref(fl)))
        return float(ms.value), float(fl.value)

    def sigma_bytes(self) -> float:
        out = C.c_double()
        self._check(self._lib.sqd_sigma_bytes(self._h, C.byref(out)))
        return float(out.value)

    def sigma_bytes_needed(self) -> float:
        out = C.c_double()
        self._check(self._lib.sqd_sigma_bytes_needed(self._h, C.byref(out)))
        return float(out.value)

    def sigma_kernel(self) -> str:
        """Name of the sigma kernel the current subspace selected: ``k_sigma`` (work items), ``k_sigma_direct``
        (element gather), ``k_sigma_rows<R>`` (R whole rows of C per workgroup in LDS), ``k_same_spin_mfma+k_sigma``
        (dense same-spin blocks on the f64 matrix cores + work items for the opposite-spin terms), ``k_sigma_lists``
        (large sets with short lists: link lists in registers, one pass over C and one over its transpose) or
        ``k_spmm_grouped+k_sigma`` / ``k_spmm_grouped+k_opp_rows`` (connected sets from ~10^3 strings per spin: the
        same-spin part as a sparse product in row-AXPY form on C and C^T, on groups of 8 rows -- ``k_spmm_rows``, one row
        per wavefront, beyond 32768 strings per spin; the opposite-spin terms by work items, or -- the default
        for the plain operator -- by whole rows with the beta link list in registers: ``k_opp_rows`` up to 3072 columns,
        ``k_opp_src``, passes over ranges of the source column, beyond)."""
        kind, rows = C.c_int(), C.c_int()
        self._check(self._lib.sqd_sigma_kernel(self._h, C.byref(kind), C.byref(rows)))
        spmm = "k_spmm_grouped" if rows.value > 1 else "k_spmm_rows"
        return ("k_sigma", "k_sigma_direct", f"k_sigma_rows<{rows.value}>", "k_same_spin_mfma+k_sigma",
                "k_sigma_lists", f"{spmm}+k_sigma", f"{spmm}+k_opp_rows", f"{spmm}+k_opp_src")[kind.value]

