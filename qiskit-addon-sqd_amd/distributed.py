"""Batch-parallel ``sci_solver`` for one process per GPU (``torch.distributed``; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" for CPU tests).

The reference's SQD loop hands a list of independent subspaces to ``sci_solver`` and documents that
call as its only collective step (``qiskit_addon_sqd/fermion.py:316-333``, ``:432``;
``docs/guides/hpc_acceleration.rst:52-57``); its default solver runs them serially (:670-681).
Here batch ``i`` is solved by rank ``i % world`` -- all of a rank's batches in ONE batched native solve
(``sqd_solve_batch``) -- with no communication during the solve.  Afterwards ONE all-reduce (sum of a
table whose rows are zero except on the owning rank -- i.e. a gather) makes every batch's raw observables
record known everywhere (``(5 + 2 norb) * 8`` bytes per batch: latency-bound, xGMI bandwidth irrelevant).

Round-3 exchange: the record of a batch is written INTO THE COLLECTIVE'S DEVICE BUFFER by the observables
kernel of its solve (``sqd_ctx_set_record_out``), the solver runs on torch's current stream, and the
all-reduce is enqueued behind it on the same stream -- no host-to-device copy, no host wait between solve and
exchange.  Energies and occupancies of every batch are then formed from the reduced records on every rank with
the arithmetic of the native call (``_capi.results_from_record``): the same bits everywhere.  The winner's
amplitude matrix -- the only large object the loop consumes (``fermion.py:608-631``), and only on the control
process (rank 0 does the carry-over, ``:436-451``) -- travels from its owner to rank 0 alone; nothing is
broadcast.  ``states="all"`` gathers every batch's state on rank 0 (one grouped send / receive), for callbacks
that read them (``fermion.py:435-436``).

Reference semantics (v0.13) = argmin over energies, take that batch's occupancies (``fermion.py:577,
:604-605``): ``occupancy_reduce="best"``.  ``"mean"`` replaces every result's occupancies by the batch
average (the older tutorial workflow / BASELINE north_star wording), computed from the same table.
"""

from __future__ import annotations

import time
from typing import Callable, Sequence

import numpy as np

from . import _capi
from .fermion import SCIResult, SCIState, _DeferredAmplitudes, _davidson_kwargs, _get_context, _settle_deferred, solve_sci


def _dist():
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("torch.distributed is not initialised; call init_process_group first")
    return dist


import os as _os

_HOOK_ON = _os.environ.get("SQD_DIST_HOOK", "1") != "0"  # (probe switch: the exchange enqueued by the solve's hook, or after it)
_XBUF: dict = {}
_GROUPS: dict = {}
last_exchange_ms: float | None = None
"""Wall clock of the latest call's exchange on this rank: from the return of the local solve(s) to the reduced table
on the host, plus the winner's transfer (benchmark hook; ``bench.py`` reports it as ``exchange_ms``)."""


def _wait_stream(stream) -> None:
    """Wait for ``stream``: a short polling spin (``hipStreamSynchronize`` parks the thread and wakes up 10-20 us
    late; the waits here cover a 500-byte collective), then the blocking call -- a long wait must not burn a core
    and hold the GIL."""
    t0 = time.perf_counter()
    while not stream.query():
        if time.perf_counter() - t0 > 2e-4:
            stream.synchronize()
            return


def _group_info(dist, group):
    """(rank, world, backend is RCCL) of a process group, cached: three c10d queries cost ~10 us per call."""
    key = id(group)
    hit = _GROUPS.get(key)
    if hit is None or hit[3] is not (group if group is not None else dist.group.WORLD):
        pg = group if group is not None else dist.group.WORLD
        hit = _GROUPS[key] = (dist.get_rank(group), dist.get_world_size(group), dist.get_backend(group) == "nccl", pg)
        while len(_GROUPS) > 16:
            _GROUPS.pop(next(iter(_GROUPS)))
    return hit[0], hit[1], hit[2]


class _Exchange:
    """The path's one collective as a callable the solver's enqueue hook can hold: all-reduce of the record table and
    the copy of the result to page-locked memory, enqueued on the current stream.  One object per table, reused by every
    call (a fresh closure per call is cyclic garbage, and what that costs is a full pass of Python's collector over
    everything ``import torch`` created -- 35 ms -- every thousand calls or so)."""

    __slots__ = ("dev", "host", "group", "done")

    def __init__(self, dev, host, group):
        self.dev, self.host, self.group, self.done = dev, host, group, False

    def __call__(self):
        import torch.distributed as dist

        dist.all_reduce(self.dev, op=dist.ReduceOp.SUM, group=self.group)
        self.host.copy_(self.dev, non_blocking=True)
        self.done = True


def _exchange_buffers(group, tdev, nb: int, width: int, on_gpu: bool):
    import torch

    key = (id(group), str(tdev), nb, width)
    hit = _XBUF.get(key)
    if hit is None:
        host = torch.zeros((nb, width), dtype=torch.float64)
        if on_gpu:
            host = host.pin_memory()
        dev = torch.zeros((nb, width), dtype=torch.float64, device=tdev) if on_gpu else host
        hit = _XBUF[key] = (dev, host, _Exchange(dev, host, group))
        while len(_XBUF) > 16:
            _XBUF.pop(next(iter(_XBUF)))
    return hit


def _precheck_nelec(pairs, indices, nelec):
    """Host-side check of this rank's batches against ``nelec`` (popcount of each list's first string -- the rule of
    reference ``solve_fermion``, fermion.py:797-798); returns the exception to raise behind the exchange, or None."""
    for i, (sa, sb) in zip(indices, pairs):
        try:
            got = (int(sa[0]).bit_count() if len(sa) else 0, int(sb[0]).bit_count() if len(sb) else 0)
        except Exception as exc:  # noqa: BLE001 -- malformed input: reported like any other failure of the batch
            return ValueError(f"batch {i}: cannot read the CI strings ({exc!r})")
        if len(sa) == 0 or len(sb) == 0:
            return ValueError(f"batch {i}: empty CI string list")
        if got != tuple(nelec):
            return ValueError(f"nelec={tuple(nelec)} does not match the Hamming weights {got} of the CI strings")
    return None


def shard_indices(num_batches: int, rank: int, world: int) -> list[int]:
    """Batches owned by ``rank``: round-robin, ``i % world == rank``."""
    return list(range(rank, num_batches, world))


class _RemoteAmplitudes(_DeferredAmplitudes):
    """Placeholder for the state of a batch another rank solved and that was not shipped (``states="winner"``)."""

    __slots__ = ("owner",)

    def __init__(self, index, shape, owner):
        super().__init__(None, index, shape)
        self.owner = owner

    def fetch(self):
        raise RuntimeError(
            f"the state of batch {self.index} lives on rank {self.owner}: only the lowest-energy batch is shipped to "
            "the control process; call solve_sci_batch_distributed(..., states='all') to gather every state on rank 0"
        )


def solve_sci_batch_distributed(
    ci_strings: Sequence[tuple[np.ndarray, np.ndarray]],
    one_body_tensor: np.ndarray,
    two_body_tensor: np.ndarray,
    norb: int,
    nelec: tuple[int, int],
    *,
    spin_sq: float | None = None,
    group=None,
    device: int | None = None,
    occupancy_reduce: str = "best",
    states: str = "winner",
    local_solver: Callable[..., SCIResult] | None = None,
    **kwargs,
) -> list[SCIResult]:
    """Collective drop-in for ``solve_sci_batch`` (same positional signature, so it can be passed as
    ``sci_solver=`` to the SQD loop on every rank).

    Returns one ``SCIResult`` per batch on every rank.  ``energy`` and ``orbital_occupancies`` are populated for all
    batches.  ``sci_state``: the batches this rank solved; on the control process (group rank 0) also the
    lowest-energy batch, or every batch with ``states="all"``; every other state is a placeholder that raises when
    its amplitudes are read.  ``rdm1``/``rdm2`` (lazy) only where the state is present.
    """
    global last_exchange_ms
    if occupancy_reduce not in ("best", "mean"):
        raise ValueError("occupancy_reduce must be 'best' or 'mean'")
    if states not in ("winner", "all"):
        raise ValueError("states must be 'winner' or 'all'")
    import torch

    dist = _dist()
    rank, world, on_gpu = _group_info(dist, group)
    if device is None:
        device = torch.cuda.current_device() if on_gpu else 0
    tdev = torch.device("cuda", device) if on_gpu else torch.device("cpu")
    one_body_tensor = np.asarray(one_body_tensor, dtype=np.float64)
    norb = int(one_body_tensor.shape[0])
    nb = len(ci_strings)
    width = _capi.record_width(norb)
    nelec = tuple(int(x) for x in nelec)
    mine = shard_indices(nb, rank, world)
    compute_rdms = kwargs.pop("compute_rdms", "lazy")
    shift = 0.2  # pyscf fix_spin_ default, as solve_sci (reference fermion.py:715)

    dev_table, host_table, exchange = _exchange_buffers(group, tdev, nb, width, on_gpu)
    exchange.done = False
    table = host_table.numpy()
    local: dict[int, SCIResult] = {}
    resident = None  # (context, position of each local batch in its batched solve)
    failure = None   # an exception of this rank's solves: raised behind the exchange, so that no rank is left waiting

    # ---- independent solves, no communication.  Their records land in the collective's buffer by themselves.
    if local_solver is None and compute_rdms is not True and kwargs.get("ci0") is None:
        ctx = _get_context(one_body_tensor, two_body_tensor, device, slot="dist")
        _settle_deferred(ctx)
        if on_gpu:
            stream = torch.cuda.current_stream(tdev)
            if getattr(ctx, "_on_stream", None) != stream.cuda_stream:
                ctx.use_stream(stream.cuda_stream)  # solver kernels and the collective on ONE stream
                ctx._on_stream = stream.cuda_stream
        dev_table.zero_()
        host_formed = False  # (every rank, with or without a batch of its own: the records are the kernels')
        rec_host = None
        if mine:
            import weakref

            from . import fermion as _F

            if on_gpu:
                ctx.set_record_out(dev_table.data_ptr() + 8 * rank * width, world * width)
            else:
                # a CPU backend (gloo): the collective's table is pageable host memory, which the observables kernel of a
                # real GPU must not be handed -- it writes its records into page-locked memory, copied over below
                rec_host = _capi.pinned_empty((nb, width))
                rec_host[:] = 0.0
                ctx.set_record_out(rec_host.ctypes.data + 8 * rank * width, world * width)
            if on_gpu and _HOOK_ON:
                # the exchange is enqueued by the solve itself, right behind its last kernel (before its final host
                # wait): the stream does not idle while this thread gets back from the native call and into RCCL
                ctx.set_enqueue_hook(exchange)
            dk = _davidson_kwargs(kwargs)
            dk.pop("verbose", None)
            dk.pop("ci0", None)
            # What can be checked about this rank's batches is checked BEFORE the solve: once the solve's hook has
            # enqueued the exchange, a failure can no longer be announced through it.  The native build
            # validates order and a common Hamming weight per list before its first launch, i.e. also before the hook.
            failure = _precheck_nelec([ci_strings[i] for i in mine], mine, nelec)
            try:
                if failure is not None:
                    out = None
                elif len(mine) == 1:
                    # one batch per rank (BASELINE config 3: 8 batches over 8 GPUs): the single solve.  The control
                    # process takes the state with the call (written by the observables kernel into page-locked memory);
                    # elsewhere it stays on the device -- it only ever leaves for rank 0, GPU to GPU
                    sa, sb = ci_strings[mine[0]]
                    amps1, st1, (e1, _s2, oa1, ob1) = ctx.solve(sa, sb, spin_sq=spin_sq, shift=shift, spin_square=False,
                                                                fetch=(rank == 0), **dk)
                    out = {"nelec": [ctx.nelec], "energy": [e1], "occ_a": [oa1], "occ_b": [ob1], "stats": [st1],
                           "amps": [amps1]}
                else:
                    out = ctx.solve_batch([ci_strings[i] for i in mine], spin_sq=spin_sq, shift=shift, spin_square=False,
                                          fetch="none", **dk)
            except Exception as exc:  # noqa: BLE001 -- re-raised behind the exchange
                failure, out = exc, None
            finally:
                ctx.set_record_out(None)
                if on_gpu and _HOOK_ON:
                    ctx.set_enqueue_hook(None)
            if failure is None:
                try:
                    ctx.raise_hook_error()
                except Exception as exc:  # noqa: BLE001
                    failure, out = exc, None
            if failure is None:
                for k in range(len(mine)):
                    if out["nelec"][k] != nelec:
                        failure = ValueError(f"nelec={nelec} does not match the Hamming weights {out['nelec'][k]} of the CI strings")
                        break
            if rec_host is not None and failure is None:
                table[mine, :] = rec_host[mine, :]
            if failure is not None:
                # poison this rank's records: every rank sees the NaN behind the all-reduce and raises with this one
                if on_gpu:
                    dev_table[mine, 0] = float("nan")
                else:
                    table[mine, 0] = np.nan
            else:
                resident = (ctx, {i: k for k, i in enumerate(mine)}, len(mine) == 1)
                _F._TLS.stats, _F._TLS.batch_stats = out["stats"][0], out["stats"]
            for k, i in enumerate(mine if failure is None else []):
                sa, sb = ci_strings[i]
                amps = out["amps"][k]
                if amps is None:
                    amps = _DeferredAmplitudes(ctx, k, (len(sa), len(sb)), out.get("generation"))
                    if len(mine) == 1:
                        amps.single = True
                    ctx._deferred.append(weakref.ref(amps))
                state = SCIState(amps, np.asarray(sa), np.asarray(sb), norb=norb, nelec=nelec)
                local[i] = SCIResult._make(float(out["energy"][k]), state, (out["occ_a"][k], out["occ_b"][k]),
                                           lazy=(compute_rdms == "lazy"))
    else:
        # a caller-supplied solver (or eager RDMs / a start vector): the records are formed on the host
        solver = local_solver or solve_sci
        host_formed = True
        table[:] = 0.0
        for i in mine:
            try:
                res = solver(ci_strings[i], one_body_tensor, two_body_tensor, norb=norb, nelec=nelec, spin_sq=spin_sq,
                             device=device, compute_rdms=compute_rdms, **kwargs)  # fmt: skip
            except Exception as exc:  # noqa: BLE001 -- re-raised behind the exchange
                failure = exc
                table[i, 0] = np.nan
                break
            local[i] = res
            # a record that results_from_record maps back to exactly these numbers: c.c = 1, no penalty left to apply
            table[i, 0] = res.energy
            table[i, 3] = 1.0
            table[i, 4 : 4 + norb] = res.orbital_occupancies[0]
            table[i, 4 + norb : 4 + 2 * norb] = res.orbital_occupancies[1]
        if on_gpu:
            dev_table.copy_(host_table, non_blocking=True)

    # ---- the path's single exchange: all-reduce(sum) of the per-batch records, enqueued behind the solves
    t_x = time.perf_counter()
    if on_gpu:
        if not exchange.done:  # (no local batch, a caller-supplied solver, or the hook switched off)
            exchange()
        _wait_stream(torch.cuda.current_stream(tdev))
    else:
        dist.all_reduce(host_table, op=dist.ReduceOp.SUM, group=group)
    table = table.copy()
    if failure is not None:
        raise failure
    # A record is unusable when its owner poisoned it (NaN energy: the owner failed before the exchange was enqueued) or
    # when the kernels themselves produced no state (c.c not a positive finite number: the owner's solve_collect raises
    # "zero norm" AFTER its hook has enqueued the exchange, so the record is the only messenger).  Every rank applies
    # the same test to the same reduced table: all of them raise, none is left waiting in the state transfer below.
    cc = table[:, 3]
    unusable = np.isnan(table[:, 0]) | ~np.isfinite(cc) | ~(cc > 0.0)
    if unusable.any():
        bad = [int(i) for i in np.flatnonzero(unusable)]
        raise RuntimeError(f"solve_sci_batch_distributed: the solves of batches {bad} failed on ranks "
                           f"{sorted({i % world for i in bad})} (their own exceptions are raised there)")
    energies = np.empty(nb)
    occs = []
    for i in range(nb):
        e, oa, ob = _capi.results_from_record(table[i], norb, None if host_formed else spin_sq, shift, nelec)
        energies[i] = e
        occs.append((oa, ob))
    best = int(np.argmin(energies))

    # ---- states to the control process (group rank 0): the winner's, or all of them
    wanted = list(range(nb)) if states == "all" else [best]
    shipped: dict[int, np.ndarray] = {}
    moves = [(i, i % world) for i in wanted if i % world != 0]
    if moves and world > 1:
        ops, bufs = [], {}
        g0 = dist.get_global_rank(group, 0) if group is not None else 0
        for i, owner in moves:
            shape = (len(ci_strings[i][0]), len(ci_strings[i][1]))
            if rank == owner:
                if resident is not None and on_gpu:
                    holder = resident[0] if resident[2] else resident[0].batch_sub(resident[1][i])
                    sub_ptr = holder.solution_device_ptr()
                    ta = torch.as_tensor(_DeviceView(sub_ptr, shape), device=tdev)
                else:
                    ta = torch.from_numpy(np.ascontiguousarray(local[i].sci_state.amplitudes, dtype=np.float64))
                    if on_gpu:
                        ta = ta.to(tdev, non_blocking=True)
                ops.append(dist.P2POp(dist.isend, ta, g0, group=group))
                bufs[i] = ta
            elif rank == 0:
                gsrc = dist.get_global_rank(group, owner) if group is not None else owner
                ta = torch.empty(shape, dtype=torch.float64, device=tdev)
                ops.append(dist.P2POp(dist.irecv, ta, gsrc, group=group))
                bufs[i] = ta
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()  # (RCCL: the CURRENT STREAM waits for the transfer, not this thread -- what follows is ordered)
            if rank == 0:
                for i, ta in bufs.items():
                    if on_gpu:
                        amps = _capi.pinned_empty(tuple(ta.shape))
                        torch.from_numpy(amps).copy_(ta, non_blocking=True)
                        shipped[i] = amps
                    else:
                        shipped[i] = ta.numpy()
                if on_gpu:
                    _wait_stream(torch.cuda.current_stream(tdev))
    last_exchange_ms = 1e3 * (time.perf_counter() - t_x)

    if occupancy_reduce == "mean":
        mean_occ = (np.mean([o[0] for o in occs], axis=0), np.mean([o[1] for o in occs], axis=0))
    out_list: list[SCIResult] = []
    for i in range(nb):
        occ = mean_occ if occupancy_reduce == "mean" else occs[i]
        sa, sb = ci_strings[i]
        if i in local:
            r = local[i]
            raw = r.__dict__.get  # (do not trigger lazy RDMs)
            out_list.append(SCIResult._make(float(energies[i]), r.sci_state, occ, rdm1=raw("rdm1"), rdm2=raw("rdm2"),
                                            lazy=r._is_lazy()))
        elif i in shipped:
            state = SCIState(shipped[i], np.asarray(sa), np.asarray(sb), norb=norb, nelec=nelec)
            out_list.append(SCIResult._make(float(energies[i]), state, occ, lazy=True))
        else:
            state = SCIState(_RemoteAmplitudes(i, (len(sa), len(sb)), i % world), np.asarray(sa), np.asarray(sb),
                             norb=norb, nelec=nelec)
            out_list.append(SCIResult(float(energies[i]), state, occ))
    return out_list


class _DeviceView:
    """Zero-copy handle on device memory for ``torch.as_tensor`` (``__cuda_array_interface__``, version 2)."""

    def __init__(self, ptr: int, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
