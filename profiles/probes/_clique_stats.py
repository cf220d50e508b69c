"""CPU analysis (not a test): the (N-2)-electron cliques of the same-spin double links of HF-centred sets -- what a
clique-by-clique dense product on the matrix cores could cover (profiles/r06/clique_stats.txt)."""
import sys, itertools, collections
import numpy as np
sys.path.insert(0, "/root/repo")
from qiskit_addon_sqd_amd import synthetic as S
norb, nel = 30, 8
for n in (1000, 3000, 5000):
    strs = S.hf_centred_strings(norb, nel, n, 11)
    idx = {int(s): i for i, s in enumerate(strs)}
    cl = collections.defaultdict(list)
    for i, s in enumerate(strs):
        s = int(s)
        occ = [k for k in range(norb) if s >> k & 1]
        for p, q in itertools.combinations(occ, 2):
            cl[s & ~(1 << p) & ~(1 << q)].append(i)
    sizes = np.array([len(v) for v in cl.values()])
    # double links: pairs in a clique whose strings differ by 4 bits (pairs sharing an orbital differ by 2 bits: singles)
    tot_d = 0
    per = {}
    for K, mem in cl.items():
        m = len(mem)
        if m < 2: continue
        ss = np.array([int(strs[i]) for i in mem], dtype=np.uint64)
        x = ss[:, None] ^ ss[None, :]
        pc = np.zeros(x.shape, int)
        for b in range(norb): pc += ((x >> np.uint64(b)) & np.uint64(1)).astype(int)
        nd = int((pc == 4).sum())
        per[K] = (m, nd)
        tot_d += nd
    print(f"n={n} cliques={len(cl)} mean size={sizes.mean():.2f} max={sizes.max()} double links/string={tot_d/n:.1f}")
    for mmin in (8, 12, 16, 24, 32, 48):
        big = [(m, nd) for (m, nd) in per.values() if m >= mmin]
        cov = sum(nd for m, nd in big)
        prow = sum(m for m, nd in big)
        pad = sum(((m + 15) // 16 * 16) * ((m + 3) // 4 * 4) for m, nd in big)
        full = sum(m * m for m, nd in big)
        print(f"   m>={mmin:3d}: cliques={len(big):6d} covered double links={cov/max(tot_d,1):.3f} P rows/string={prow/n:.2f} "
              f"dense m^2/covered={full/max(cov,1):.2f} padded/covered={pad/max(cov,1):.2f}")
