"""numpy oracle (O1) for the fermionic subspace-diagonalization hot path.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Never imported by the
product package.

Three layers, each independent of the HIP implementation:

* integer layer  -- restates ``qiskit_addon_sqd`` reference code line by line
  (bit-exact; pinned by the reference's literals, ``tests/golden``);
* brute force    -- Jordan-Wigner matrices of the second-quantised Hamiltonian
  and S^2 on the full Fock space (norb <= 7), projected onto the determinant
  subset.  This is the anchor: it shares no formula with anything else;
* Slater-Condon  -- string-space construction of P H P for larger subspaces
  (checked against the brute force in ``tests/test_oracle.py``).

Floating-point layer: **parity unpinned against pyscf** (absent here); the
conventions follow SURVEY.md Appendix A (pyscf>=2.9 ``fci.selected_ci``).

Conventions (reference ``fermion.py:1027-1035``):
 * a CI string is an integer; bit p = occupation of spatial orbital p (LSB = 0);
 * basis |A>|B> = alpha string (right half of a bitstring) x beta string;
 * amplitudes ``C[ia, ib]`` row-major, flat index ``ia*nb + ib``;
 * ``eri[p,q,r,s] = (pq|rs)`` chemist order, real, 8-fold symmetric;
 * ``H = sum h_pq a+_p a_q + 1/2 sum (pq|rs) a+_p a+_r a_s a_q``.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------
# integer layer (restates the reference; bit-exact)
# --------------------------------------------------------------------------


def bitstring_matrix_to_integers(bitstring_matrix: np.ndarray) -> np.ndarray:
    """Row -> integer, column 0 = MSB.  Follows reference ``counts.py:186-201``:
    int64 accumulation when n_bits < 64, Python-object integers otherwise."""
    n_bitstrings, n_bits = bitstring_matrix.shape
    if n_bits < 64:
        result = np.zeros(n_bitstrings, dtype=int)
        mat = bitstring_matrix
    else:
        result = np.zeros(n_bitstrings, dtype=object)
        mat = bitstring_matrix.astype(object)
    for i in range(n_bits):
        result += mat[:, i] * (1 << (n_bits - 1 - i))
    return result


def bitstring_matrix_to_ci_strs(bitstring_matrix: np.ndarray, open_shell: bool = False):
    """Follows reference ``fermion.py:1004-1035``: left half = beta, right half =
    alpha, unique+sorted per spin, closed shell => union for both; returns (alpha, beta)."""
    norb = bitstring_matrix.shape[1] // 2
    left = np.unique(bitstring_matrix_to_integers(bitstring_matrix[:, :norb]))
    right = np.unique(bitstring_matrix_to_integers(bitstring_matrix[:, norb:]))
    if not open_shell:
        left = right = np.union1d(left, right)
    return right, left


def check_ci_strs(ci_strs):
    """Follows reference ``fermion.py:1075-1097`` (same messages)."""
    addr_up, addr_dn = ci_strs
    ham0 = format(addr_up[0], "b").count("1")
    for i, addr in enumerate(addr_up):
        ham = format(addr, "b").count("1")
        if ham != ham0:
            raise ValueError(
                f"Spin-up CI string in index 0 has hamming weight {ham0}, but CI string in "
                f"index {i} has hamming weight {ham}."
            )
    ham0 = format(addr_dn[0], "b").count("1")
    for i, addr in enumerate(addr_dn):
        ham = format(addr, "b").count("1")
        if ham != ham0:
            raise ValueError(
                f"Spin-down CI string in index 0 has hamming weight {ham0}, but CI string in "
                f"index {i} has hamming weight {ham}."
            )
    return np.sort(np.unique(addr_up)), np.sort(np.unique(addr_dn))


# --------------------------------------------------------------------------
# brute force: Jordan-Wigner on the full Fock space
# --------------------------------------------------------------------------


def jw_annihilators(nmodes: int):
    """Sparse annihilation operators a_j, j < nmodes, on the 2**nmodes Fock space.
    Basis state index n: bit j = occupation of mode j.  a_j|n> = (-1)^{sum_{k<j} n_k}|n - e_j>."""
    dim = 1 << nmodes
    idx = np.arange(dim, dtype=np.int64)
    ops = []
    for j in range(nmodes):
        occ = (idx >> j) & 1
        cols = idx[occ == 1]
        rows = cols ^ (1 << j)
        below = cols & ((1 << j) - 1)
        sign = 1.0 - 2.0 * (np.bitwise_count(below.astype(np.uint64)) & 1)
        ops.append(sp.csr_matrix((sign, (rows, cols)), shape=(dim, dim)))
    return ops


def jw_hamiltonian(h1: np.ndarray, eri: np.ndarray):
    """Full-space H with modes [alpha 0..norb-1, beta 0..norb-1]."""
    norb = h1.shape[0]
    a = jw_annihilators(2 * norb)
    ad = [x.T.tocsr() for x in a]
    dim = 1 << (2 * norb)
    H = sp.csr_matrix((dim, dim))
    for s in (0, norb):
        for p in range(norb):
            for q in range(norb):
                if h1[p, q] != 0.0:
                    H = H + h1[p, q] * (ad[s + p] @ a[s + q])
    for s in (0, norb):
        for t in (0, norb):
            for q in range(norb):
                for sidx in range(norb):
                    right = a[t + sidx] @ a[s + q]
                    if right.nnz == 0:
                        continue
                    for p in range(norb):
                        for r in range(norb):
                            v = eri[p, q, r, sidx]
                            if v != 0.0:
                                H = H + (0.5 * v) * (ad[s + p] @ ad[t + r] @ right)
    return H.tocsr()


def jw_spin_square(norb: int):
    a = jw_annihilators(2 * norb)
    ad = [x.T.tocsr() for x in a]
    dim = 1 << (2 * norb)
    splus = sp.csr_matrix((dim, dim))
    sz = sp.csr_matrix((dim, dim))
    for p in range(norb):
        splus = splus + ad[p] @ a[norb + p]
        sz = sz + 0.5 * (ad[p] @ a[p] - ad[norb + p] @ a[norb + p])
    return (splus.T @ splus + sz + sz @ sz).tocsr()


def jw_basis_indices(strs_a, strs_b, norb: int) -> np.ndarray:
    """Full-space index of |A>|B>, flattened ia*nb+ib."""
    A = np.asarray(strs_a, dtype=np.int64)[:, None]
    B = np.asarray(strs_b, dtype=np.int64)[None, :]
    return (A | (B << norb)).ravel()


def jw_project(op, strs_a, strs_b, norb: int) -> np.ndarray:
    idx = jw_basis_indices(strs_a, strs_b, norb)
    return np.asarray(op[idx][:, idx].todense())


def jw_rdms(c: np.ndarray, strs_a, strs_b, norb: int):
    """(dm1a, dm1b, dm2) by definition; dm2[p,q,r,s] = sum_{st} <p+_s r+_t s_t q_s>
    (pyscf convention)."""
    a = jw_annihilators(2 * norb)
    ad = [x.T.tocsr() for x in a]
    dim = 1 << (2 * norb)
    psi = np.zeros(dim)
    psi[jw_basis_indices(strs_a, strs_b, norb)] = np.asarray(c).ravel()
    dm1 = np.zeros((2, norb, norb))
    for si, s in enumerate((0, norb)):
        for p in range(norb):
            for q in range(norb):
                dm1[si, p, q] = psi @ (ad[s + p] @ (a[s + q] @ psi))
    dm2 = np.zeros((norb,) * 4)
    for s in (0, norb):
        for t in (0, norb):
            for q in range(norb):
                for sidx in range(norb):
                    v = a[t + sidx] @ (a[s + q] @ psi)
                    if not np.any(v):
                        continue
                    for p in range(norb):
                        for r in range(norb):
                            dm2[p, q, r, sidx] += psi @ (ad[s + p] @ (ad[t + r] @ v))
    return dm1[0], dm1[1], dm2


def jw_rdm2s(c: np.ndarray, strs_a, strs_b, norb: int):
    """(dm2aa, dm2ab, dm2bb) by definition, pyscf ``make_rdm2s`` convention (reference ``fermion.py:124-125``):
    dm2st[p,q,r,s] = <p+_s r+_t s_t q_s>; the spin-summed dm2 is dm2aa + dm2bb + dm2ab + dm2ab^T(2,3,0,1)."""
    a = jw_annihilators(2 * norb)
    ad = [x.T.tocsr() for x in a]
    psi = np.zeros(1 << (2 * norb))
    psi[jw_basis_indices(strs_a, strs_b, norb)] = np.asarray(c).ravel()
    out = []
    for s, t in ((0, 0), (0, norb), (norb, norb)):
        dm2 = np.zeros((norb,) * 4)
        for q in range(norb):
            for sidx in range(norb):
                v = a[t + sidx] @ (a[s + q] @ psi)
                if not np.any(v):
                    continue
                for p in range(norb):
                    for r in range(norb):
                        dm2[p, q, r, sidx] = psi @ (ad[s + p] @ (ad[t + r] @ v))
        out.append(dm2)
    return tuple(out)


# --------------------------------------------------------------------------
# string-space links (canonical order: by target address, then source address)
# --------------------------------------------------------------------------


def _u64(x):
    return np.asarray(x, dtype=np.uint64)


def _bit(p):
    return np.uint64(1) << np.uint64(p)


def pair_index(p, q):
    """pyscf 'tril' pair index p(p+1)/2+q for p>=q."""
    hi = np.maximum(p, q)
    lo = np.minimum(p, q)
    return hi * (hi + 1) // 2 + lo


def single_links(strs, norb: int):
    """All in-set single excitations.  Returns dict of arrays (tgt, src, p, q, sign):
    |strs[tgt]> = sign * a+_p a_q |strs[src]>, p != q; sorted by (tgt, src).
    This is the off-diagonal part of pyscf's ``cre_des_linkstr_tril`` table
    (SURVEY A.3 / row a8), in this build's canonical order."""
    strs = _u64(strs)
    n = len(strs)
    out = {k: [] for k in ("tgt", "src", "p", "q", "sign")}
    for q in range(norb):
        has_q = (strs >> np.uint64(q)) & np.uint64(1)
        for p in range(norb):
            if p == q:
                continue
            has_p = (strs >> np.uint64(p)) & np.uint64(1)
            src = np.nonzero((has_q == 1) & (has_p == 0))[0]
            if src.size == 0:
                continue
            J = strs[src]
            I = (J ^ _bit(q)) | _bit(p)
            pos = np.searchsorted(strs, I)
            ok = pos < n
            ok[ok] = strs[pos[ok]] == I[ok]
            if not ok.any():
                continue
            lo, hi = min(p, q), max(p, q)
            between = ((np.uint64(1) << np.uint64(hi)) - np.uint64(1)) & ~(
                (np.uint64(1) << np.uint64(lo + 1)) - np.uint64(1)
            )
            sgn = 1 - 2 * (np.bitwise_count(J[ok] & between).astype(np.int64) & 1)
            out["tgt"].append(pos[ok])
            out["src"].append(src[ok])
            out["p"].append(np.full(ok.sum(), p))
            out["q"].append(np.full(ok.sum(), q))
            out["sign"].append(sgn)
    if not out["tgt"]:
        return {k: np.zeros(0, dtype=np.int64) for k in out}
    res = {k: np.concatenate(v).astype(np.int64) for k, v in out.items()}
    order = np.lexsort((res["src"], res["tgt"]))
    return {k: v[order] for k, v in res.items()}


def _two_bits(x):
    """x has exactly two set bits: return (high index, low index)."""
    low = x & (~x + np.uint64(1))
    high = x ^ low
    lo = np.bitwise_count(low - np.uint64(1)).astype(np.int64)
    hi = np.bitwise_count(high - np.uint64(1)).astype(np.int64)
    return hi, lo


def _apply_sign(state, orb, sgn):
    below = state & ((np.uint64(1) << orb.astype(np.uint64)) - np.uint64(1))
    return sgn * (1 - 2 * (np.bitwise_count(below).astype(np.int64) & 1))


def double_links(strs, norb: int):
    """All in-set same-spin double excitations.  Arrays (tgt, src, p, r, q, s, sign):
    |tgt> = sign * a+_p a+_r a_s a_q |src>, p>r created, q>s annihilated; sorted by (tgt, src)."""
    strs = _u64(strs)
    n = len(strs)
    keys = ("tgt", "src", "p", "r", "q", "s", "sign")
    if n == 0:
        return {k: np.zeros(0, dtype=np.int64) for k in keys}
    X = strs[:, None] ^ strs[None, :]
    tgt, src = np.nonzero(np.bitwise_count(X) == 4)
    if tgt.size == 0:
        return {k: np.zeros(0, dtype=np.int64) for k in keys}
    I, J = strs[tgt], strs[src]
    x = I ^ J
    p, r = _two_bits(x & I)
    q, s = _two_bits(x & J)
    sgn = np.ones(len(tgt), dtype=np.int64)
    state = J.copy()
    sgn = _apply_sign(state, q, sgn)
    state = state ^ (np.uint64(1) << q.astype(np.uint64))
    sgn = _apply_sign(state, s, sgn)
    state = state ^ (np.uint64(1) << s.astype(np.uint64))
    sgn = _apply_sign(state, r, sgn)
    state = state | (np.uint64(1) << r.astype(np.uint64))
    sgn = _apply_sign(state, p, sgn)
    res = dict(tgt=tgt, src=src, p=p, r=r, q=q, s=s, sign=sgn)
    return {k: np.asarray(v, dtype=np.int64) for k, v in res.items()}


def occupation_matrix(strs, norb: int) -> np.ndarray:
    strs = _u64(strs)
    return ((strs[:, None] >> np.arange(norb, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(
        np.float64
    )


# --------------------------------------------------------------------------
# Slater-Condon construction of P H P
# --------------------------------------------------------------------------


def same_spin_hamiltonian(h1, eri, strs, norb: int) -> np.ndarray:
    """Dense n x n matrix of  sum h_pq E_pq + 1/2 sum (pq|rs) a+_p a+_r a_s a_q  (one spin)
    over the string set (Slater-Condon rules)."""
    n = len(strs)
    occ = occupation_matrix(strs, norb)
    Jm = np.einsum("iijj->ij", eri)
    Km = np.einsum("ijji->ij", eri)
    H = np.zeros((n, n))
    diag = occ @ np.diag(h1) + 0.5 * np.einsum("ni,ij,nj->n", occ, Jm - Km, occ)
    H[np.arange(n), np.arange(n)] = diag
    sl = single_links(strs, norb)
    if sl["tgt"].size:
        p, q, src = sl["p"], sl["q"], sl["src"]
        # sum_k in src: (pq|kk) - (pk|kq)
        coul = eri[p, q][:, np.arange(norb), np.arange(norb)]  # [l, k] = (pq|kk)
        exch = eri[p[:, None], np.arange(norb)[None, :], np.arange(norb)[None, :], q[:, None]]  # (pk|kq)
        val = h1[p, q] + np.einsum("lk,lk->l", coul - exch, occ[src])
        np.add.at(H, (sl["tgt"], src), sl["sign"] * val)
    dl = double_links(strs, norb)
    if dl["tgt"].size:
        p, r, q, s = dl["p"], dl["r"], dl["q"], dl["s"]
        val = eri[p, q, r, s] - eri[p, s, r, q]
        np.add.at(H, (dl["tgt"], dl["src"]), dl["sign"] * val)
    return H


def excitation_operators(strs, norb: int):
    """E[p][q]: sparse n x n matrix of a+_p a_q restricted to the string set."""
    n = len(strs)
    occ = occupation_matrix(strs, norb)
    sl = single_links(strs, norb)
    E = [[None] * norb for _ in range(norb)]
    for p in range(norb):
        for q in range(norb):
            if p == q:
                E[p][q] = sp.diags(occ[:, p]).tocsr()
            else:
                m = (sl["p"] == p) & (sl["q"] == q)
                E[p][q] = sp.csr_matrix(
                    (sl["sign"][m].astype(float), (sl["tgt"][m], sl["src"][m])), shape=(n, n)
                )
    return E


def build_php(h1, eri, strs_a, strs_b, norb: int, sparse: bool = False):
    """P H P in the product basis (index ia*nb+ib) from string-space pieces:
    H = Ha (x) 1 + 1 (x) Hb + sum_{pq,rs} (pq|rs) Ea_pq (x) Eb_rs."""
    na, nb = len(strs_a), len(strs_b)
    Ha = sp.csr_matrix(same_spin_hamiltonian(h1, eri, strs_a, norb))
    Hb = sp.csr_matrix(same_spin_hamiltonian(h1, eri, strs_b, norb))
    H = sp.kron(Ha, sp.identity(nb), format="csr") + sp.kron(sp.identity(na), Hb, format="csr")
    Ea = excitation_operators(strs_a, norb)
    Eb = excitation_operators(strs_b, norb)
    for p in range(norb):
        for q in range(norb):
            if Ea[p][q].nnz == 0:
                continue
            G = sp.csr_matrix((nb, nb))
            for r in range(norb):
                for s in range(norb):
                    if Eb[r][s].nnz and eri[p, q, r, s] != 0.0:
                        G = G + eri[p, q, r, s] * Eb[r][s]
            H = H + sp.kron(Ea[p][q], G, format="csr")
    return H.tocsr() if sparse else np.asarray(H.todense())


def sigma_string_space(h1, eri, strs_a, strs_b, c, norb: int) -> np.ndarray:
    """sigma = (P H P) c at sizes where the D x D matrix of ``build_php`` cannot be formed (BASELINE's
    317 x 317 and 707 x 707): the same decomposition, H = Ha (x) 1 + 1 (x) Hb + sum_{pq,rs} (pq|rs)
    Ea_pq (x) Eb_rs, evaluated factor by factor on the amplitude matrix,
        sigma = Ha C + C Hb^T + sum_rs [sum_pq (pq|rs) Ea_pq] C Eb_rs^T,
    with dense same-spin matrices and one small dense product per beta orbital pair (r, s).  The E
    operators include their diagonal (occupation) part, so nothing here uses the J-table / hdiag split of
    the HIP kernels.  Checked against ``build_php`` (itself against the Jordan-Wigner brute force) in
    tests/test_oracle.py."""
    na, nb = len(strs_a), len(strs_b)
    C = np.asarray(c, dtype=float).reshape(na, nb)
    Ha = same_spin_hamiltonian(h1, eri, strs_a, norb)
    Hb = same_spin_hamiltonian(h1, eri, strs_b, norb)
    out = Ha @ C + C @ Hb.T

    def all_links(strs):
        """(tgt, src, p, q, sign) of every non-zero <tgt|E_pq|src>, the diagonal p = q included."""
        sl = single_links(strs, norb)
        occ = occupation_matrix(strs, norb)
        I, k = np.nonzero(occ)
        return (np.concatenate((sl["tgt"], I)), np.concatenate((sl["src"], I)), np.concatenate((sl["p"], k)),
                np.concatenate((sl["q"], k)), np.concatenate((sl["sign"], np.ones(len(I), dtype=np.int64))))

    ta, sa, pa, qa, ga = all_links(strs_a)
    tb, sb, pb, qb, gb = all_links(strs_b)
    key = pb * norb + qb
    order = np.argsort(key, kind="stable")
    bounds = np.flatnonzero(np.diff(key[order])) + 1
    for idx in np.split(order, bounds):
        r, s = int(pb[idx[0]]), int(qb[idx[0]])
        Ga = np.zeros((na, na))
        np.add.at(Ga, (ta, sa), ga * eri[pa, qa, r, s])
        # for a fixed (r, s) every source string has at most one target: plain fancy-index accumulation
        out[:, tb[idx]] += Ga @ (C[:, sb[idx]] * gb[idx])
    return out


def double_links_chunked(strs, norb: int, rows=None, chunk: int = 512):
    """``double_links`` without the n x n XOR matrix: targets are taken ``chunk`` at a time (all strings, or only the
    addresses in ``rows``), so 10^4 strings per spin (BASELINE config 2 read literally) cost 40 MB of scratch instead
    of 800 MB.  Same arrays, same canonical order (target, then source)."""
    strs = _u64(strs)
    tg = np.arange(len(strs), dtype=np.int64) if rows is None else np.asarray(rows, dtype=np.int64)
    keys = ("tgt", "src", "p", "r", "q", "s", "sign")
    parts = {k: [] for k in keys}
    for i0 in range(0, len(tg), chunk):
        t = tg[i0:i0 + chunk]
        X = strs[t][:, None] ^ strs[None, :]
        ti, src = np.nonzero(np.bitwise_count(X) == 4)
        if ti.size == 0:
            continue
        tgt = t[ti]
        I, J = strs[tgt], strs[src]
        x = I ^ J
        p, r = _two_bits(x & I)
        q, s = _two_bits(x & J)
        sgn = np.ones(len(tgt), dtype=np.int64)
        state = J.copy()
        sgn = _apply_sign(state, q, sgn)
        state = state ^ (np.uint64(1) << q.astype(np.uint64))
        sgn = _apply_sign(state, s, sgn)
        state = state ^ (np.uint64(1) << s.astype(np.uint64))
        sgn = _apply_sign(state, r, sgn)
        state = state | (np.uint64(1) << r.astype(np.uint64))
        sgn = _apply_sign(state, p, sgn)
        for k, v in zip(keys, (tgt, src, p, r, q, s, sgn)):
            parts[k].append(np.asarray(v, dtype=np.int64))
    if not parts["tgt"]:
        return {k: np.zeros(0, dtype=np.int64) for k in keys}
    return {k: np.concatenate(v) for k, v in parts.items()}


def same_spin_hamiltonian_sparse(h1, eri, strs, norb: int, rows=None):
    """``same_spin_hamiltonian`` as a scipy CSR matrix (all rows, or zero outside ``rows``): Slater-Condon diagonal,
    singles and doubles of one spin over the string set, for sets too large for the dense n x n form."""
    strs = _u64(strs)
    n = len(strs)
    occ = occupation_matrix(strs, norb)
    Jm = np.einsum("iijj->ij", eri)
    Km = np.einsum("ijji->ij", eri)
    diag = occ @ np.diag(h1) + 0.5 * np.einsum("ni,ij,nj->n", occ, Jm - Km, occ)
    keep = np.ones(n, dtype=bool)
    if rows is not None:
        keep[:] = False
        keep[np.asarray(rows, dtype=np.int64)] = True
    I = [np.flatnonzero(keep)]
    Jc = [np.flatnonzero(keep)]
    V = [diag[keep]]
    sl = single_links(strs, norb)
    if sl["tgt"].size:
        m = keep[sl["tgt"]]
        p, q, src = sl["p"][m], sl["q"][m], sl["src"][m]
        ar = np.arange(norb)
        coul = eri[p, q][:, ar, ar]
        exch = eri[p[:, None], ar[None, :], ar[None, :], q[:, None]]
        val = h1[p, q] + np.einsum("lk,lk->l", coul - exch, occ[src])
        I.append(sl["tgt"][m])
        Jc.append(src)
        V.append(sl["sign"][m] * val)
    dl = double_links_chunked(strs, norb, rows=rows)
    if dl["tgt"].size:
        p, r, q, s = dl["p"], dl["r"], dl["q"], dl["s"]
        I.append(dl["tgt"])
        Jc.append(dl["src"])
        V.append(dl["sign"] * (eri[p, q, r, s] - eri[p, s, r, q]))
    return sp.csr_matrix((np.concatenate(V), (np.concatenate(I), np.concatenate(Jc))), shape=(n, n))


def sigma_rows_string_space(h1, eri, strs_a, strs_b, c, norb: int, rows) -> np.ndarray:
    """Rows ``rows`` (alpha addresses) of ``sigma_string_space`` -- the same decomposition
        sigma = Ha C + C Hb^T + sum_{pq,rs} (pq|rs) Ea_pq C Eb_rs^T
    (E operators with their diagonal; nothing of the kernels' J-table / hdiag split) -- for subspaces where the whole
    sigma is out of a numpy oracle's reach: D = 10^8 (BASELINE config 2 read literally, 10^4 strings per spin).  The
    same-spin factors are sparse (``same_spin_hamiltonian_sparse``); the opposite-spin term is evaluated link by link
    of the requested alpha rows against ALL beta links.  Returns len(rows) x nb.  Checked against
    ``sigma_string_space`` in tests/test_oracle.py."""
    na, nb = len(strs_a), len(strs_b)
    C = np.asarray(c, dtype=float).reshape(na, nb)
    rows = np.asarray(rows, dtype=np.int64)
    Ha = same_spin_hamiltonian_sparse(h1, eri, strs_a, norb, rows=rows)
    Hb = same_spin_hamiltonian_sparse(h1, eri, strs_b, norb)
    out = (Ha[rows] @ C) + (Hb @ C[rows].T).T

    def all_links(strs, only=None):
        sl = single_links(strs, norb)
        occ = occupation_matrix(strs, norb)
        I, k = np.nonzero(occ)
        t = np.concatenate((sl["tgt"], I))
        links = (t, np.concatenate((sl["src"], I)), np.concatenate((sl["p"], k)), np.concatenate((sl["q"], k)),
                 np.concatenate((sl["sign"], np.ones(len(I), dtype=np.int64))))
        if only is None:
            return links
        m = np.isin(t, only)
        return tuple(v[m] for v in links)

    ta, sa, pa, qa, ga = all_links(strs_a, only=rows)
    tb, sb, pb, qb, gb = all_links(strs_b)
    where = {int(A): i for i, A in enumerate(rows)}
    for l in range(len(ta)):
        # <A|Ea_pq|A'> = ga:  sigma[A, B] += ga * sum_{beta links (B <- B', rs, gb)} (pq|rs) gb C[A', B']
        w = eri[pa[l], qa[l]][pb, qb] * gb * C[sa[l], sb]
        out[where[int(ta[l])]] += ga[l] * np.bincount(tb, weights=w, minlength=nb)
    return out


class StringSpaceOperator:
    """``sigma_string_space`` with the string-space pieces built once: v -> (P H P) v for the oracle's
    Davidson at BASELINE sizes (flat vectors in, flat vectors out)."""

    def __init__(self, h1, eri, strs_a, strs_b, norb: int):
        self.args = (h1, eri, strs_a, strs_b)
        self.norb = norb
        self.shape = (len(strs_a), len(strs_b))

    def __call__(self, v):
        h1, eri, sa, sb = self.args
        return sigma_string_space(h1, eri, sa, sb, np.asarray(v).reshape(self.shape), self.norb).ravel()


def bitstring_matrix_from_strings(strs_a, strs_b, norb: int) -> np.ndarray:
    """Bool sample matrix whose row i is |beta_i alpha_i> in the reference's layout (``fermion.py:1027-1035``:
    left half = spin-down, right half = spin-up, column 0 = most significant bit).  len(strs_a) == len(strs_b)."""
    a = _u64(strs_a)[:, None]
    b = _u64(strs_b)[:, None]
    shifts = np.arange(norb - 1, -1, -1, dtype=np.uint64)[None, :]
    return np.concatenate((((b >> shifts) & np.uint64(1)).astype(bool), ((a >> shifts) & np.uint64(1)).astype(bool)), axis=1)


def build_spin_square(strs_a, strs_b, norb: int, nelec, sparse: bool = False):
    """P S^2 P:  S^2 = Sz(Sz+1) + sum_p n_pb (1 - n_pa) - sum_{p!=q} Ea_qp (x) Eb_pq."""
    na, nb = len(strs_a), len(strs_b)
    sz = 0.5 * (nelec[0] - nelec[1])
    occ_a = occupation_matrix(strs_a, norb)
    occ_b = occupation_matrix(strs_b, norb)
    diag = sz * (sz + 1.0) + np.einsum("bp,ap->ab", occ_b, 1.0 - occ_a).ravel()
    S = sp.diags(diag).tocsr()
    Ea = excitation_operators(strs_a, norb)
    Eb = excitation_operators(strs_b, norb)
    for p in range(norb):
        for q in range(norb):
            if p != q and Ea[q][p].nnz and Eb[p][q].nnz:
                S = S - sp.kron(Ea[q][p], Eb[p][q], format="csr")
    return S.tocsr() if sparse else np.asarray(S.todense())


def make_hdiag(h1, eri, strs_a, strs_b, norb: int) -> np.ndarray:
    """Diagonal of P H P, formula of SURVEY row a9 (pyscf ``make_hdiag``)."""
    occ_a = occupation_matrix(strs_a, norb)
    occ_b = occupation_matrix(strs_b, norb)
    Jm = np.einsum("iijj->ij", eri)
    Km = np.einsum("ijji->ij", eri)
    ha = occ_a @ np.diag(h1) + 0.5 * np.einsum("ni,ij,nj->n", occ_a, Jm - Km, occ_a)
    hb = occ_b @ np.diag(h1) + 0.5 * np.einsum("ni,ij,nj->n", occ_b, Jm - Km, occ_b)
    return ha[:, None] + hb[None, :] + occ_a @ Jm @ occ_b.T


# --------------------------------------------------------------------------
# observables from an amplitude matrix (string-space; valid at any size the
# link enumeration can handle)
# --------------------------------------------------------------------------


def make_rdm1s(c, strs_a, strs_b, norb: int):
    """dm1a[p,q] = <a+_pa a_qa>, dm1b likewise (pyscf ``make_rdm1s`` convention)."""
    c = np.asarray(c, dtype=float)
    res = []
    for spin, strs in enumerate((strs_a, strs_b)):
        M = c if spin == 0 else c.T  # rows = strings of this spin
        occ = occupation_matrix(strs, norb)
        w = np.einsum("ij,ij->i", M, M)
        dm = np.diag(w @ occ)
        sl = single_links(strs, norb)
        if sl["tgt"].size:
            ov = np.einsum("lj,lj->l", M[sl["tgt"]], M[sl["src"]]) * sl["sign"]
            np.add.at(dm, (sl["p"], sl["q"]), ov)
        res.append(dm)
    return res[0], res[1]


def expectation(op, c) -> float:
    v = np.asarray(c, dtype=float).ravel()
    return float(v @ (op @ v))


def energy_from_rdms(h1, eri, dm1, dm2) -> float:
    """Reference ``fermion.py:730-732,827``."""
    return float(np.einsum("pr,pr->", dm1, h1) + 0.5 * np.einsum("prqs,prqs->", dm2, eri))


def make_rdm2(c, strs_a, strs_b, norb: int) -> np.ndarray:
    """Spin-summed dm2[p,q,r,s] = sum_{st} <p+_s r+_t s_t q_s> from string-space pieces
    (any norb; uses <E_pq E_rs> - delta_qr <E_ps> with the same-spin part done by explicit
    a+a+aa matrix elements so no out-of-set intermediate is dropped)."""
    c = np.asarray(c, dtype=float)
    na, nb = c.shape
    dm2 = np.zeros((norb,) * 4)
    # opposite spin: <Ea_pq Eb_rs> + <Eb_pq Ea_rs>
    Ea = excitation_operators(strs_a, norb)
    Eb = excitation_operators(strs_b, norb)
    ab = np.zeros((norb,) * 4)
    for p in range(norb):
        for q in range(norb):
            if Ea[p][q].nnz == 0:
                continue
            left = Ea[p][q].T @ c  # (Ea_pq^T c)[A', B] -> sum_A c[A,B] Ea[A,A']
            for r in range(norb):
                for s in range(norb):
                    if Eb[r][s].nnz == 0:
                        continue
                    # sum_{A,B,A',B'} c[A,B] Ea_pq[A,A'] Eb_rs[B,B'] c[A',B']
                    ab[p, q, r, s] = np.sum(left * (c @ Eb[r][s].T))
    dm2 += ab + ab.transpose(2, 3, 0, 1)
    # same spin
    for spin, strs in enumerate((strs_a, strs_b)):
        M = c if spin == 0 else c.T
        occ = occupation_matrix(strs, norb)
        w = np.einsum("ij,ij->i", M, M)
        # diagonal: n_p n_r (p != r): +dm2[p,p,r,r], -dm2[p,r,r,p]
        nn = np.einsum("i,ip,ir->pr", w, occ, occ)
        for p in range(norb):
            for r in range(norb):
                if p != r:
                    dm2[p, p, r, r] += nn[p, r]
                    dm2[p, r, r, p] -= nn[p, r]
        sl = single_links(strs, norb)
        if sl["tgt"].size:
            ov = np.einsum("lj,lj->l", M[sl["tgt"]], M[sl["src"]]) * sl["sign"]
            for l in range(len(ov)):
                a_, b_ = sl["p"][l], sl["q"][l]
                for k in np.nonzero(occ[sl["src"][l]])[0]:
                    if k == b_:
                        continue
                    # <a+_a a+_k a_k a_b> = sign ; index forms (p,q,r,s): operator a+_p a+_r a_s a_q
                    dm2[a_, b_, k, k] += ov[l]
                    dm2[k, k, a_, b_] += ov[l]
                    dm2[a_, k, k, b_] -= ov[l]
                    dm2[k, b_, a_, k] -= ov[l]
        dl = double_links(strs, norb)
        if dl["tgt"].size:
            ov = np.einsum("lj,lj->l", M[dl["tgt"]], M[dl["src"]]) * dl["sign"]
            p, r, q, s = dl["p"], dl["r"], dl["q"], dl["s"]
            np.add.at(dm2, (p, q, r, s), ov)
            np.add.at(dm2, (r, s, p, q), ov)
            np.add.at(dm2, (p, s, r, q), -ov)
            np.add.at(dm2, (r, q, p, s), -ov)
    return dm2


# --------------------------------------------------------------------------
# Davidson (pyscf ``lib.davidson1`` control flow, single root)
# --------------------------------------------------------------------------


def davidson_pyscf(aop, x0, hdiag, tol=1e-9, lindep=1e-14, max_cycle=100, max_space=12):
    """Single-root Davidson with pyscf's control flow.  Returns (converged, e, x, n_sigma)."""
    toloose = np.sqrt(tol)
    x0 = np.asarray(x0, dtype=float)
    x0 = x0 / np.linalg.norm(x0)
    xs, axs = [], []
    xt = x0
    e = 0.0
    nsig = 0
    conv = False
    space = 0
    for _ in range(max_cycle):
        xs.append(xt)
        axs.append(aop(xt))
        nsig += 1
        space += 1
        heff = np.array([[xi @ axj for axj in axs] for xi in xs])
        heff = 0.5 * (heff + heff.T)
        w, v = np.linalg.eigh(heff)
        elast, e = e, w[0]
        v0 = v[:, 0]
        x = sum(vi * xi for vi, xi in zip(v0, xs))
        ax = sum(vi * axi for vi, axi in zip(v0, axs))
        r = ax - e * x
        rnorm = np.linalg.norm(r)
        de = e - elast if space > 1 or nsig > 1 else e
        if abs(de) < tol and rnorm < toloose:
            conv = True
            break
        if rnorm**2 <= lindep:
            conv = rnorm < toloose
            break
        t = r / (hdiag - e + 1e-4)
        t = t / np.linalg.norm(t)
        for xi in xs:
            t = t - xi * (xi @ t)
        tn = np.linalg.norm(t)
        if tn**2 <= lindep:
            conv = rnorm < toloose
            break
        xt = t / tn
        if space + 1 > max_space:
            xs, axs, space = [], [], 0
            xt = x / np.linalg.norm(x)
    return conv, e, x, nsig


def init_guess(hdiag: np.ndarray, na: int, nb: int, nelec=None) -> np.ndarray:
    """pyscf ``get_init_guess`` for selected CI: unit vector at the lowest diagonal element
    with +1e-5 / -1e-5 on the first/last element.  pyscf ``direct_spin1._get_init_guess`` (reached through
    ``SelectedCI.get_init_guess``) searches only the lower triangle ``A >= B`` (``lib.pack_tril``) when
    ``neleca == nelecb and na == nb``, so that the start vector does not favour one of two spin-mirrored
    determinants; pass ``nelec`` to apply that rule (ties go to the first element in row-major order, as
    numpy's ``argpartition``/``argmin`` on the packed triangle do for a unique minimum)."""
    h = np.asarray(hdiag, dtype=float).reshape(na, nb)
    if nelec is not None and nelec[0] == nelec[1] and na == nb:
        ia, ib = np.tril_indices(na)
        k = int(np.argmin(h[ia, ib]))
        addr = int(ia[k]) * nb + int(ib[k])
    else:
        addr = int(np.argmin(h))
    x = np.zeros(na * nb)
    x[addr] = 1.0
    x[0] += 1e-5
    x[-1] -= 1e-5
    return x


# --------------------------------------------------------------------------
# end-to-end oracle of the reference wrappers
# --------------------------------------------------------------------------


def solve_fermion_dense(bitstring_matrix, hcore, eri, open_shell=False, spin_sq=None, shift=0.1):
    """Dense restatement of reference ``solve_fermion`` (``fermion.py:745-845``): exact ground
    state of P (H + penalty) P by ``eigh`` instead of Davidson; energy, occupancies and <S^2>
    of that state.  Returns (e, amplitudes[na,nb], (occ_a, occ_b), s2, (strs_a, strs_b))."""
    if isinstance(bitstring_matrix, tuple):
        ci_strs = bitstring_matrix
    else:
        ci_strs = bitstring_matrix_to_ci_strs(bitstring_matrix, open_shell=open_shell)
    strs_a, strs_b = check_ci_strs(ci_strs)
    norb = hcore.shape[0]
    nelec = (format(int(strs_a[0]), "b").count("1"), format(int(strs_b[0]), "b").count("1"))
    H = build_php(hcore, eri, strs_a, strs_b, norb)
    S2 = build_spin_square(strs_a, strs_b, norb, nelec)
    Heff = H
    if spin_sq is not None:
        # pyscf fix_spin_
        sz = abs(nelec[0] - nelec[1]) * 0.5
        P = S2 - spin_sq * np.eye(len(S2))
        if spin_sq < sz * (sz + 1) + 0.1:
            Heff = H + shift * P
        else:
            Heff = H + shift * (P @ P)
    w, v = np.linalg.eigh(Heff)
    c = v[:, 0]
    e = float(c @ H @ c)
    amps = c.reshape(len(strs_a), len(strs_b))
    dm1a, dm1b = make_rdm1s(amps, strs_a, strs_b, norb)
    s2 = float(c @ S2 @ c)
    return e, amps, (np.diagonal(dm1a).copy(), np.diagonal(dm1b).copy()), s2, (strs_a, strs_b)


# --------------------------------------------------------------------------
# synthetic problems
# --------------------------------------------------------------------------


def synthetic_integrals(norb: int, seed: int | None = None):
    """Seeded gapped h1 and 8-fold symmetric PSD eri via density fitting."""
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


def random_strings(norb: int, nelec: int, n: int, rng) -> np.ndarray:
    """n distinct uniform-random particle-conserving strings, sorted."""
    rng = np.random.default_rng(rng)
    out = set()
    while len(out) < n:
        pos = rng.choice(norb, nelec, replace=False)
        out.add(int(sum(1 << int(p) for p in pos)))
    return np.array(sorted(out), dtype=np.int64)


def hf_centred_strings(norb: int, nelec: int, n: int, rng) -> np.ndarray:
    """n distinct strings with excitation rank k ~ Geometric(0.5) from the aufbau string."""
    rng = np.random.default_rng(rng)
    hf = (1 << nelec) - 1
    out = {hf}
    while len(out) < n:
        k = min(int(rng.geometric(0.5)), nelec, norb - nelec)
        occ = rng.choice(nelec, k, replace=False)
        vir = nelec + rng.choice(norb - nelec, k, replace=False)
        s = hf
        for o in occ:
            s ^= 1 << int(o)
        for v in vir:
            s |= 1 << int(v)
        out.add(s)
    return np.array(sorted(out), dtype=np.int64)
