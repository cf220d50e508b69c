"""CPU logic tests: the unmodified HIP kernel sources, compiled by g++ against tests/emu and run by
OS threads, checked against the numpy oracle.  These are NOT the parity tests proper (those are
``-m gpu`` in test_gpu_parity.py); they catch indexing/sign/bounds bugs without a GPU."""
import numpy as np
import pytest

from oracle import sqd_oracle as O
from qiskit_addon_sqd_amd import _capi

from _parity import check_link_tables, check_operators, make_problem, run_full_parity, run_operator_parity


@pytest.mark.parametrize(
    "norb,nelec,na,nb,seed,hf",
    [
        (6, (3, 2), 12, 9, 5, False),
        (7, (3, 3), 20, 20, 7, True),
        (5, (1, 4), 5, 4, 9, False),   # nalpha < nbeta, single alpha electron (no alpha doubles)
        (4, (2, 2), 6, 6, 3, False),   # complete space (FCI limit)
    ],
)
def test_emu_full_parity(emu_lib, norb, nelec, na, nb, seed, hf):
    # the solver-variant sweep (restarts, long basis, ...) only on the smallest case: the emulator is slow
    run_full_parity(emu_lib, norb, nelec, na, nb, seed, hf, variants=(na * nb <= 20))


def test_emu_capped_ell_overflow_rows(emu_lib, monkeypatch):
    # SQD_ELL_CAP=3 cuts the beta link lists into many overflow chunks (virtual rows): the partial sums
    # that travel through LDS must reproduce the same sigma / ground state
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)
    run_operator_parity(emu_lib, 6, (2, 3), 9, 14, 5, False)


def test_emu_many_axpy_items(emu_lib, monkeypatch):
    # SQD_SIGMA_L=2 cuts the same-spin alpha links into many 2-link AXPY items per row (partial rows + the
    # fixed-order reduce with many slots per row)
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_L", "2")
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    run_operator_parity(emu_lib, 8, (4, 4), 9, 30, 5, False)


@pytest.mark.parametrize("direct", ["0", "1"])
def test_emu_direct_and_work_item_sigma(emu_lib, monkeypatch, direct):
    # SQD_SIGMA_DIRECT forces (1) / forbids (0) the element-gather sigma kernel that ultra-sparse string sets take
    # by default: both kernels must reproduce the oracle on the same inputs (H, S^2, both penalty forms, Davidson)
    monkeypatch.setenv("SQD_SIGMA_DIRECT", direct)
    run_full_parity(emu_lib, 6, (3, 2), 12, 9, 5, False, variants=False)
    run_operator_parity(emu_lib, 7, (3, 3), 20, 20, 7, True)


def test_emu_fused_direct_sigma_and_reduction(emu_lib, monkeypatch):
    # element-gather formulation inside a Davidson run: the sigma build and the fused dot products / eigen step are ONE
    # launch (k_sigma_dots_eig, round 6) -- against the two launches (SQD_DAV_FUSE_DIRECT=0; what batched solves and timed
    # runs keep) bit for bit: plain operator and the linear spin penalty, bases of more than 8 vectors, restarts
    # (max_space 4), a start vector from the caller
    monkeypatch.setenv("SQD_SIGMA_DIRECT", "1")
    h1, eri, sa, sb = make_problem(8, (4, 3), 40, 30, 13)
    ci0 = np.random.default_rng(5).standard_normal((len(sa), len(sb)))
    runs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SQD_DAV_FUSE_DIRECT", fused)
        with _capi.Context(h1, eri, lib=emu_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == "k_sigma_direct"
            out = []
            # (time_sigma_every=2: every other sigma launch goes out alone between its events, the rest fused)
            for kw in ({}, {"spin_sq": 0.75, "shift": 0.3}, {"max_space": 4}, {"ci0": ci0, "tol": 1e-11},
                       {"time_sigma_every": 2}):
                c, st = ctx.davidson(**kw)
                assert st["converged"], kw
                # (two workgroups per BLAS-1 launch at D = 1200: the solution is formed by the launch that stops the solve,
                # and the workgroup that starts behind the one raising the flag has to do its share)
                assert abs(np.linalg.norm(c) - 1.0) < 1e-12, kw
                out.append((c.copy(), st["e_davidson"], st["n_sigma"], st["iterations"]))
            runs[fused] = out
    for a, b in zip(runs["1"], runs["0"]):
        assert np.array_equal(a[0], b[0]) and a[1:] == b[1:]
    assert max(r[2] for r in runs["1"]) > 9  # (the thirteen-slot loop ran)


@pytest.mark.parametrize("rows", ["1", "2", "3", "8"])
def test_emu_rows_kernel(emu_lib, monkeypatch, rows):
    # SQD_SIGMA_ROWS=R forces k_sigma_rows (R whole rows of C per workgroup in LDS, beta doubles in per-slice
    # jagged-diagonal order), which large uniform-random sets take by default: ragged last workgroup (12 rows in
    # groups of 8), ragged last slice (70 columns), all operator forms, Davidson
    monkeypatch.setenv("SQD_SIGMA_ROWS", rows)
    h1, eri, sa, sb = make_problem(8, (3, 4), 12, 70, 23)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == f"k_sigma_rows<{rows}>"
    run_full_parity(emu_lib, 8, (3, 4), 12, 70, 23, False, variants=False)
    run_operator_parity(emu_lib, 7, (3, 3), 20, 20, 7, True)
    run_operator_parity(emu_lib, 9, (2, 4), 7, 100, 29, True)


def test_emu_global_row_fallback(emu_lib, monkeypatch):
    # SQD_SIGMA_GLOBAL_ROWS=64 forces the path taken when a C row does not fit LDS: rows are read in
    # place, one alpha link per batch, and the beta side is cut into 64-column chunks (here 2 chunks,
    # the second ragged) with their own virtual-row ranges
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_GLOBAL_ROWS", "64")
    run_full_parity(emu_lib, 8, (3, 4), 12, 70, 23, False, variants=False)
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    run_operator_parity(emu_lib, 9, (2, 4), 7, 100, 29, True)


def test_emu_multi_pass_partial_sums(emu_lib, monkeypatch):
    # SQD_SIGMA_PASS=5 leaves LDS room for only 5 partial sums per list: the virtual rows of the staged
    # row are walked in several passes (the layout of rows with ~10^4 strings), singles and doubles needing
    # different numbers of passes
    monkeypatch.setenv("SQD_SIGMA_DENSE", "0")  # (the sparse same-spin work items are what this test is about)
    monkeypatch.setenv("SQD_SIGMA_PASS", "5")
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)
    # 420 / 840 virtual rows in passes of 40, two column strides per thread (64 threads: the emulator's
    # barriers are OS-thread rendezvous)
    monkeypatch.setenv("SQD_SIGMA_PASS", "40")
    monkeypatch.setenv("SQD_ELL_CAP", "3")
    monkeypatch.setenv("SQD_SIGMA_T", "64")
    run_operator_parity(emu_lib, 8, (3, 4), 12, 70, 23, False)


def test_emu_h2_minimal(emu_lib):
    # H2 / STO-3G textbook integrals: 2 electrons in 2 orbitals, 2x2 subspace
    h1 = np.diag([-1.2525, -0.4759])
    eri = np.zeros((2, 2, 2, 2))
    eri[0, 0, 0, 0], eri[1, 1, 1, 1] = 0.6746, 0.6974
    eri[0, 0, 1, 1] = eri[1, 1, 0, 0] = 0.6636
    for p, q, r, s in [(0, 1, 0, 1), (0, 1, 1, 0), (1, 0, 0, 1), (1, 0, 1, 0)]:
        eri[p, q, r, s] = 0.1813
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace([1, 2], [1, 2])
        amps, st = ctx.davidson()
        e = ctx.energy()
    H = O.jw_project(O.jw_hamiltonian(h1, eri), [1, 2], [1, 2], 2)
    assert abs(e - np.linalg.eigvalsh(H)[0]) < 1e-10
    assert abs(e + 0.7137 - (-1.1373)) < 1e-3  # literature total energy, 4 digits


def test_emu_one_by_one(emu_lib):
    h1, eri, sa, sb = make_problem(5, (2, 2), 1, 1, 4)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace(sa, sb)
        amps, st = ctx.davidson()
        assert st["converged"] == 1 and amps.shape == (1, 1) and abs(abs(amps[0, 0]) - 1) < 1e-12
        assert abs(ctx.energy() - O.make_hdiag(h1, eri, sa, sb, 5)[0, 0]) < 1e-12
        ctx.use_stream(0)  # sqd_ctx_use_stream: adopt a caller-owned stream (the null stream here), solve again
        ctx.set_subspace(sa, sb)
        ctx.davidson()
        assert abs(ctx.energy() - O.make_hdiag(h1, eri, sa, sb, 5)[0, 0]) < 1e-12


@pytest.mark.parametrize(
    "norb,nelec,na,nb,seed",
    [
        (4, (2, 0), 4, 1, 1),   # no beta electrons: the beta string table is the single string 0
        (4, (0, 2), 1, 5, 2),   # no alpha electrons
        (3, (3, 3), 1, 1, 4),   # every orbital doubly occupied
        (4, (4, 1), 1, 3, 5),   # full alpha shell, one beta electron
        (6, (1, 0), 6, 1, 6),   # one electron in total
    ],
)
def test_emu_empty_and_full_shells(emu_lib, norb, nelec, na, nb, seed):
    h1, eri, sa, sb = make_problem(norb, nelec, na, nb, seed)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace(sa, sb)
        ctx.davidson()
        e, s2, (da, db) = ctx.energy(), ctx.spin_square(), ctx.rdm1s()
    H = O.jw_project(O.jw_hamiltonian(h1, eri), sa, sb, norb)
    assert abs(e - np.linalg.eigvalsh(H)[0]) < 1e-10
    assert abs(np.trace(da) - nelec[0]) < 1e-10 and abs(np.trace(db) - nelec[1]) < 1e-10
    sz = 0.5 * (nelec[0] - nelec[1])
    assert s2 > sz * (sz + 1) - 1e-9


def test_emu_ragged_widths(emu_lib):
    # nb not a multiple of 64 and > 64: exercises sliced-ELL slice boundaries and partial waves
    norb, nelec = 9, (2, 3)
    h1, eri, sa, sb = make_problem(norb, nelec, 7, 70, 21)
    rng = np.random.default_rng(0)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace(sa, sb)
        check_link_tables(ctx, sa, sb, norb, h1, eri)
        check_operators(ctx, h1, eri, sa, sb, norb, nelec, rng)


def test_emu_invalid_inputs(emu_lib):
    h1, eri, sa, sb = make_problem(6, (3, 2), 5, 4, 1)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        bad = sa.copy()
        bad[2] = 0b1111  # Hamming weight 4 instead of 3
        with pytest.raises(ValueError, match="hamming weight"):
            ctx.set_subspace(np.sort(bad), sb)
        with pytest.raises(ValueError, match="strictly ascending"):
            ctx.set_subspace(sa[::-1].copy(), sb)
        with pytest.raises(ValueError, match="empty"):
            ctx.set_subspace(np.array([], dtype=np.int64), sb)
        with pytest.raises(_capi.SQDNativeError, match="no subspace"):
            ctx.hdiag()


def test_emu_dense_same_spin_mfma(emu_lib, monkeypatch):
    # SQD_SIGMA_DENSE=1 forces the mode connected string sets take by default (same-spin blocks >= 8 % dense): the
    # same-spin part H_a C + C H_b as one dense product on the f64 matrix cores (v_mfma_f64_16x16x4_f64, emulated
    # lane for lane here), the work items keeping the opposite-spin terms.  Ragged tiles (70 and 20 against 64), all
    # operator forms, Davidson, RDMs
    monkeypatch.setenv("SQD_SIGMA_DENSE", "1")
    monkeypatch.setenv("SQD_SIGMA_DIRECT", "0")
    h1, eri, sa, sb = make_problem(7, (3, 3), 20, 20, 7, True)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_same_spin_mfma+k_sigma"
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)
    run_operator_parity(emu_lib, 9, (4, 2), 70, 30, 31, True)
    monkeypatch.delenv("SQD_SIGMA_DENSE")
    monkeypatch.delenv("SQD_SIGMA_DIRECT")
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:  # default selection: 20 HF-centred strings of 7 orbitals are dense
        ctx.set_subspace(sa, sb)
        assert ctx.sigma_kernel() == "k_same_spin_mfma+k_sigma"


def test_emu_enqueue_hook(emu_lib):
    """sqd_ctx_set_enqueue_hook: called once per solve call (single and batched), between the last launch and the final
    wait; removable; an exception raised inside does not cross the C frames and is handed back afterwards."""
    h1, eri, sa, sb = make_problem(6, (3, 2), 8, 7, 3)
    _, _, sa2, sb2 = make_problem(6, (3, 2), 6, 9, 4)
    with _capi.Context(h1, eri, lib=emu_lib) as ctx:
        calls = []
        ctx.set_enqueue_hook(lambda: calls.append(1))
        _amps, _st, (e1, *_rest) = ctx.solve(sa, sb, spin_square=False)
        assert len(calls) == 1
        out = ctx.solve_batch([(sa, sb), (sa2, sb2)], spin_square=False, fetch="all")
        assert len(calls) == 2 and abs(out["energy"][0] - e1) < 1e-12
        ctx.set_enqueue_hook(None)
        ctx.solve(sa, sb, spin_square=False)
        assert len(calls) == 2

        def boom():
            raise RuntimeError("from the hook")

        ctx.set_enqueue_hook(boom)
        _amps, _st, (e2, *_rest) = ctx.solve(sa, sb, spin_square=False)  # the solve itself completes
        assert abs(e2 - e1) < 1e-12
        ctx.set_enqueue_hook(None)
        with pytest.raises(RuntimeError, match="from the hook"):
            ctx.raise_hook_error()
        ctx.raise_hook_error()  # (raised once)


def test_emu_lists_kernel(emu_lib, monkeypatch):
    # SQD_SIGMA_LISTS=1 forces the list passes (sqd_lists.hip: link lists in registers, rows of C / C^T through LDS,
    # transpose + compact single x single kernel) that 10^4 x 10^4 sets take by default.  Small cases reach every
    # branch: lists longer than the registers' 16 links and more than 4 single links per string (tails in the LDS
    # overflow tables: 30 of the 70 strings of (4e,8o) have ~22 links, ~7 of them singles), ragged row chunks, odd row
    # lengths (unaligned tile stores), nalpha != nbeta, all operator forms, a Davidson solve and the observables
    monkeypatch.setenv("SQD_SIGMA_LISTS", "1")

    def selected(norb, nelec, na, nb, seed, hf):
        h1, eri, sa, sb = make_problem(norb, nelec, na, nb, seed, hf)
        with _capi.Context(h1, eri, lib=emu_lib) as ctx:
            ctx.set_subspace(sa, sb)
            return ctx.sigma_kernel() == "k_sigma_lists", ctx.link_counts(0), ctx.link_counts(1)

    cases = [(7, (3, 3), 20, 20, 7, True), (6, (2, 3), 9, 14, 5, False), (6, (3, 2), 13, 9, 5, False),
             (5, (1, 4), 5, 4, 9, False), (8, (4, 4), 30, 28, 17, False),
             (16, (4, 4), 66, 70, 23, False)]  # (the last one: full 64 x 64 tiles in the 16-byte transpose path)
    for case in cases:
        ok, la, lb = selected(*case)
        assert ok, case
    _, la, lb = selected(*cases[-1])
    assert la[0] + la[1] > 16 * 30 and la[0] > 4 * 30  # the overflow tables are really in use
    run_full_parity(emu_lib, *cases[0], variants=False)
    for case in cases[1:]:
        run_operator_parity(emu_lib, *case)
    # the alpha side by rows (k_alpha_rows, the default) holds a list one link per lane, 64 at a time; with the chunk cut
    # to 3 links every list of these cases is walked in several reloads (the path lists longer than 64 links take)
    monkeypatch.setenv("SQD_ALPHA_CHUNK", "3")
    run_operator_parity(emu_lib, *cases[1])
    run_operator_parity(emu_lib, *cases[4])
    monkeypatch.delenv("SQD_ALPHA_CHUNK")
    # a set whose tails exceed the overflow tables (complete beta space: 261 links per string) is refused: another kernel
    ok, _, _ = selected(12, (2, 6), 3, 300, 19, False)
    assert not ok


@pytest.mark.parametrize("xcd,J,form,opp", [("1", "1", "g", "1"), ("0", "1", "g", "0"), ("0", "2", "r", "1"), ("1", "4", "r", "0"),
                                            ("1", "2", "g", "1")])
def test_emu_spmm_same_spin(emu_lib, monkeypatch, xcd, J, form, opp):
    # SQD_SIGMA_SPMM=1 forces the sparse-product same-spin path (sqd_spmm.hip: C -> C^T, the product on C and C^T, G +=
    # G2T^T) that connected sets from ~900 strings per spin take by default.
    #  form = g: the DEFAULT product (k_spmm_grouped: 8 adjacent rows share the sorted union of their source lists, a dense
    #            8-vector of coefficients per source; one and two columns per lane); r: k_spmm_rows on the merged lists
    #            (what more than 32 768 strings per spin take)
    #  opp = 1:  the opposite-spin part and the diagonal by whole rows (sqd_opp.hip: beta links in registers, entries
    #            staged interleaved, per-link sums folded to columns, long rows in pieces with partial rows) for the plain
    #            operator and the linear spin penalty -- the default.  opp = 0: work items throughout (they add ONE partial
    #            product).
    # Both task mappings (XCD split on / off), every panel width, ragged panels / groups, nalpha != nbeta, rows without
    # links, all operator forms, a Davidson solve and the observables.
    monkeypatch.setenv("SQD_SIGMA_SPMM", "1")
    monkeypatch.setenv("SQD_SIGMA_OPP", opp)
    monkeypatch.setenv("SQD_SPMM_XCD", xcd)
    monkeypatch.setenv("SQD_SPMM_J", J)
    monkeypatch.setenv("SQD_SPMM_GJ", J if J in ("1", "2") else "1")
    monkeypatch.setenv("SQD_SPMM_GROUPED", "1" if form == "g" else "0")
    monkeypatch.setenv("SQD_OPP_E", "4")  # (pieces of 4 entries: rows in several pieces, partial rows, the deferred sum)
    tiled = form
    default = (tiled, opp) == ("g", "1")
    cases = [(7, (3, 3), 20, 20, 7, True), (6, (2, 3), 9, 14, 5, False), (5, (1, 4), 5, 4, 9, False)]
    if default:
        cases += [(8, (4, 4), 30, 28, 17, True), (9, (2, 4), 7, 100, 29, True), (16, (4, 4), 66, 70, 23, True)]
    if form == "g":  # more than one panel, a ragged last group
        cases += [(12, (3, 3), 140, 24, 31, True), (12, (3, 3), 18, 150, 33, True)]
    for case in cases:
        h1, eri, sa, sb = make_problem(*case)
        with _capi.Context(h1, eri, lib=emu_lib) as ctx:
            ctx.set_subspace(sa, sb)
            spmm = "k_spmm_grouped" if form == "g" else "k_spmm_rows"
            assert ctx.sigma_kernel() == (f"{spmm}+k_opp_rows" if opp == "1" else f"{spmm}+k_sigma"), case
    run_full_parity(emu_lib, *cases[0], variants=False)
    for case in cases[1:]:
        run_operator_parity(emu_lib, *case)


def test_emu_opp_rows_column_ranges(emu_lib, monkeypatch):
    # the whole-row opposite-spin kernel with SEVERAL column ranges per row (what 3000 strings per spin take: the beta
    # link list no longer fits one workgroup's registers) and columns whose links span several threads: 64-thread
    # workgroups holding 1 link per thread; pieces of 4 entries (partial rows, the deferred sum inside the Davidson run)
    monkeypatch.setenv("SQD_SIGMA_SPMM", "1")
    monkeypatch.setenv("SQD_OPP_T", "64")
    monkeypatch.setenv("SQD_OPP_S", "1")
    monkeypatch.setenv("SQD_OPP_E", "4")
    for case in ((7, (3, 3), 20, 20, 7, True), (8, (4, 4), 30, 28, 17, True), (9, (2, 4), 7, 100, 29, True)):
        h1, eri, sa, sb = make_problem(*case)
        with _capi.Context(h1, eri, lib=emu_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == "k_spmm_grouped+k_opp_rows" and ctx.link_counts(1)[0] > 64
        run_operator_parity(emu_lib, *case)
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)


@pytest.mark.parametrize("hooks", [{"SQD_OPPS_T": "64", "SQD_OPPS_S": "4"}, {"SQD_OPPS_T": "64", "SQD_OPPS_S": "8", "SQD_OPPS_E": "4"},
                                   {"SQD_OPPS_T": "128", "SQD_OPPS_BIG": "1", "SQD_OPPS_E": "10"}])
def test_emu_opp_src_passes(emu_lib, monkeypatch, hooks):
    # k_opp_src (sqd_oppsrc.hip: what rows of more than 3072 columns take): passes over ranges of the SOURCE column, the
    # links of a range grouped by excitation operator in sub-runs of four, a piece's entries staged eight at a time, the
    # per-link sums through LDS to the owners of the target columns.  64-thread workgroups on small sets: many passes,
    # ragged ranges, pieces of more than eight entries (several staging rounds) and of four (partial rows + the deferred
    # sum inside the Davidson run), one and two sub-runs per thread, both LDS layouts, rows of 2 .. 8 columns per thread,
    # the linear spin penalty (run_operator_parity), a whole solve
    monkeypatch.setenv("SQD_SIGMA_SPMM", "1")
    monkeypatch.setenv("SQD_OPP_SRC", "1")
    for k, v in hooks.items():
        monkeypatch.setenv(k, v)
    cases = [(7, (3, 3), 20, 20, 7, True), (9, (2, 4), 7, 100, 29, True)]
    if "SQD_OPPS_BIG" in hooks:
        cases += [(8, (4, 4), 30, 28, 17, True)]
    else:
        cases += [(11, (2, 5), 6, 230, 29, True)] if hooks["SQD_OPPS_S"] == "8" else [(11, (3, 5), 5, 460, 33, True)]
    for case in cases:
        h1, eri, sa, sb = make_problem(*case)
        with _capi.Context(h1, eri, lib=emu_lib) as ctx:
            ctx.set_subspace(sa, sb)
            assert ctx.sigma_kernel() == "k_spmm_grouped+k_opp_src", (case, ctx.sigma_kernel())
        run_operator_parity(emu_lib, *case)
    run_full_parity(emu_lib, 7, (3, 3), 20, 20, 7, True, variants=False)


def test_emu_rdm2_opposite_spin_row_form(emu_lib, monkeypatch):
    # SQD_RDM2_ROWS=64 forces the row form of the opposite-spin block of rdm2 (sqd_rdm.hip: both extended link lists sorted
    # by orbital pair on the device, an alpha chunk's rows staged in LDS, the beta list in registers) that sets with 2e8
    # and more link pairs take by default -- here with 64-thread workgroups: one and several beta ranges, chunks of one
    # pair cut at 32 links, against the oracle's rdm2 and against the thread-per-link form ("0")
    from oracle import sqd_oracle as O

    for case in ((7, (3, 3), 20, 20, 7, True), (11, (2, 5), 6, 300, 31, True), (10, (4, 2), 120, 9, 3, False)):
        norb = case[0]
        h1, eri, sa, sb = make_problem(*case)
        amps = np.random.default_rng(case[4]).standard_normal((len(sa), len(sb)))
        amps /= np.linalg.norm(amps)
        out = {}
        for hook in ("64", "0"):
            monkeypatch.setenv("SQD_RDM2_ROWS", hook)
            with _capi.Context(h1, eri, lib=emu_lib) as ctx:
                ctx.set_subspace(sa, sb)
                out[hook] = ctx.rdm2(amps)
                if hook == "64":
                    aa, ab, bb = ctx.rdm2s(amps)
        ref = O.make_rdm2(amps, sa, sb, norb)
        assert np.allclose(out["64"], ref, atol=1e-12) and np.allclose(out["0"], ref, atol=1e-12), case
        # resolved form: dm2 = aa + bb + ab + ab^T(2,3,0,1)
        assert np.allclose(aa + bb + ab + ab.transpose(2, 3, 0, 1), ref, atol=1e-12), case
