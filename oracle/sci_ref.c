/* oracle/sci_ref.c -- CPU restatement (O2) of the pyscf selected-CI solve that the reference calls.
 *
 * TEST INFRASTRUCTURE ONLY (checker + `cpu_baseline` of bench.py); never loaded by the product.
 *
 * The reference (qiskit_addon_sqd/fermion.py:721-723, :810-818) delegates to
 * pyscf.fci.selected_ci.kernel_fixed_space (pyscf>=2.9, reference pyproject.toml:30; NOT vendored in
 * /root/reference and not installed here => PARITY UNPINNED against pyscf itself).  This file restates
 * pyscf's published algorithm as summarised in SURVEY.md Appendix A -- same data structures
 * (cre_des / des_des link tables, front-packed, zero-sign terminated), same dense formulation
 * (gather t1, dgemm with the packed integrals, scatter), same Davidson control flow -- and is
 * validated against the independent brute-force oracle in oracle/sqd_oracle.py (tests/test_oracle.py).
 *
 *   A.2 absorb_h1e          -> ref_absorb_h1e
 *   A.3 link tables         -> ref_cre_des_linkstr_tril, ref_des_uniq_strs, ref_des_des_linkstr_tril
 *   A.4 contract_2e         -> ref_contract_2e  (SCIcontract_2e_aaaa x2 + SCIcontract_2e_bbaa)
 *   A.5 make_hdiag          -> ref_make_hdiag
 *   A.6 davidson1           -> ref_davidson
 *   A.7 rdm1 diagonal       -> ref_occupancies
 *
 * dgemm: an OpenBLAS cblas_dgemm entry can be injected at run time (ref_set_dgemm, resolved by
 * oracle/sci_ref.py from the OpenBLAS bundled with numpy/scipy); otherwise a plain blocked loop is used.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef void (*dgemm_fn)(int order, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a,
                         int64_t lda, const double* b, int64_t ldb, double beta, double* c, int64_t ldc);
static dgemm_fn g_dgemm = NULL;
void ref_set_dgemm(void* fn) { g_dgemm = (dgemm_fn)fn; }
void ref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* C[m,n] = A[m,k] * B[k,n], row major */
static void gemm_nn(int64_t m, int64_t n, int64_t k, const double* a, const double* b, double* c) {
  if (g_dgemm) {
    g_dgemm(101 /*RowMajor*/, 111, 111, m, n, k, 1.0, a, k, b, n, 0.0, c, n);
    return;
  }
  memset(c, 0, sizeof(double) * m * n);
  for (int64_t i = 0; i < m; ++i)
    for (int64_t l = 0; l < k; ++l) {
      const double av = a[i * k + l];
      if (av == 0.0) continue;
      const double* br = b + l * n;
      double* cr = c + i * n;
      for (int64_t j = 0; j < n; ++j) cr[j] += av * br[j];
    }
}

static int64_t find_str(const uint64_t* strs, int64_t n, uint64_t s) {
  int64_t lo = 0, hi = n - 1;
  while (lo <= hi) {
    int64_t mid = (lo + hi) / 2;
    if (strs[mid] == s) return mid;
    if (strs[mid] < s) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}
static int popc(uint64_t x) { return __builtin_popcountll(x); }
static uint64_t below(int p) { return p >= 64 ? ~0ull : ((1ull << p) - 1ull); }
/* sign of a+_p a_q on string s (q occupied, p empty or p == q) */
static int cre_des_sign(int p, int q, uint64_t s) {
  if (p == q) return 1;
  int lo = p < q ? p : q, hi = p < q ? q : p;
  uint64_t between = below(hi) & ~below(lo + 1);
  return (popc(s & between) & 1) ? -1 : 1;
}

/* A.3: cre_des table, tril pair index p(p+1)/2+q (p>=q).  link[nstrs][nlink][4] = {pair, 0, addr, sign},
 * nlink = nocc + nocc*nvir; diagonal entries first, then in-set single excitations; zero sign terminates. */
void ref_cre_des_linkstr_tril(int32_t* link, int norb, int64_t nstrs, int nocc, const uint64_t* strs) {
  const int nvir = norb - nocc;
  const int nlink = nocc + nocc * nvir;
  memset(link, 0, sizeof(int32_t) * nstrs * nlink * 4);
#pragma omp parallel for schedule(static)
  for (int64_t id = 0; id < nstrs; ++id) {
    int occ[64], vir[64], no = 0, nv = 0;
    const uint64_t s0 = strs[id];
    for (int i = 0; i < norb; ++i) {
      if ((s0 >> i) & 1ull) occ[no++] = i; else vir[nv++] = i;
    }
    int32_t* tab = link + id * nlink * 4;
    int k = 0;
    for (; k < no; ++k) {
      tab[k * 4 + 0] = occ[k] * (occ[k] + 1) / 2 + occ[k];
      tab[k * 4 + 2] = (int32_t)id;
      tab[k * 4 + 3] = 1;
    }
    for (int a = 0; a < nv; ++a)
      for (int i = 0; i < no; ++i) {
        const uint64_t s1 = (s0 ^ (1ull << occ[i])) | (1ull << vir[a]);
        const int64_t addr = find_str(strs, nstrs, s1);
        if (addr < 0) continue;
        const int p = vir[a], q = occ[i];
        tab[k * 4 + 0] = p > q ? p * (p + 1) / 2 + q : q * (q + 1) / 2 + p;
        tab[k * 4 + 2] = (int32_t)addr;
        tab[k * 4 + 3] = cre_des_sign(p, q, s0);
        ++k;
      }
  }
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y);
}
/* A.3: sorted unique set of all strings with two electrons removed.  Returns count; out sized nstrs*nocc*(nocc-1)/2 */
int64_t ref_des_uniq_strs(uint64_t* out, int norb, int64_t nstrs, int nocc, const uint64_t* strs) {
  int64_t n = 0;
  for (int64_t id = 0; id < nstrs; ++id) {
    int occ[64], no = 0;
    for (int i = 0; i < norb; ++i)
      if ((strs[id] >> i) & 1ull) occ[no++] = i;
    for (int i = 0; i < no; ++i)
      for (int j = 0; j < i; ++j) out[n++] = strs[id] ^ (1ull << occ[i]) ^ (1ull << occ[j]);
  }
  qsort(out, n, sizeof(uint64_t), cmp_u64);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i)
    if (i == 0 || out[i] != out[m - 1]) out[m++] = out[i];
  return m;
}
/* A.3: des_des table over intermediates: for every creator pair p>r empty in k with k|p|r in the set:
 * {p(p-1)/2+r, 0, addr, sign of a+_p a+_r on k}.  Row capacity nlink = (nvir+2)*(nvir+1)/2. */
void ref_des_des_linkstr_tril(int32_t* link, int norb, int64_t nstrs, int nocc, const uint64_t* strs,
                              int64_t ninter, const uint64_t* inter, int nlink) {
  memset(link, 0, sizeof(int32_t) * ninter * nlink * 4);
#pragma omp parallel for schedule(static)
  for (int64_t id = 0; id < ninter; ++id) {
    const uint64_t k0 = inter[id];
    int32_t* tab = link + id * nlink * 4;
    int k = 0;
    for (int p = 1; p < norb; ++p) {
      if ((k0 >> p) & 1ull) continue;
      for (int r = 0; r < p; ++r) {
        if ((k0 >> r) & 1ull) continue;
        const uint64_t s1 = k0 | (1ull << p) | (1ull << r);
        const int64_t addr = find_str(strs, nstrs, s1);
        if (addr < 0) continue;
        /* a+_r first (acts on k0), then a+_p (acts on k0|r) */
        int par = popc(k0 & below(r)) + popc((k0 | (1ull << r)) & below(p));
        tab[k * 4 + 0] = p * (p - 1) / 2 + r;
        tab[k * 4 + 2] = (int32_t)addr;
        tab[k * 4 + 3] = (par & 1) ? -1 : 1;
        ++k;
      }
    }
  }
}

/* A.5 */
void ref_make_hdiag(double* hdiag, const double* h1, const double* eri, int norb, int64_t na, int64_t nb,
                    const uint64_t* sa, const uint64_t* sb) {
  const int64_t n1 = norb, n2 = n1 * n1, n3 = n2 * n1;
#pragma omp parallel for schedule(static)
  for (int64_t ia = 0; ia < na; ++ia) {
    for (int64_t ib = 0; ib < nb; ++ib) {
      double e = 0.0;
      for (int i = 0; i < norb; ++i) {
        const int oa = (sa[ia] >> i) & 1, ob = (sb[ib] >> i) & 1;
        if (!oa && !ob) continue;
        e += (oa + ob) * h1[i * n1 + i];
        for (int j = 0; j < norb; ++j) {
          const int pa = (sa[ia] >> j) & 1, pb = (sb[ib] >> j) & 1;
          const double J = eri[i * n3 + i * n2 + j * n1 + j], K = eri[i * n3 + j * n2 + j * n1 + i];
          e += 0.5 * ((oa * pa + ob * pb) * (J - K) + (oa * pb + ob * pa) * J);
        }
      }
      hdiag[ia * nb + ib] = e;
    }
  }
}

/* A.2: h2e = (eri with f folded in) * fac */
void ref_absorb_h1e(double* h2e, const double* h1, const double* eri, int norb, int nelec, double fac) {
  const int64_t n1 = norb, n2 = n1 * n1, n3 = n2 * n1, n4 = n3 * n1;
  double* f = (double*)malloc(sizeof(double) * n2);
  for (int p = 0; p < norb; ++p)
    for (int q = 0; q < norb; ++q) {
      double s = 0.0;
      for (int i = 0; i < norb; ++i) s += eri[p * n3 + i * n2 + i * n1 + q];
      f[p * n1 + q] = (h1[p * n1 + q] - 0.5 * s) / (nelec + 1e-100);
    }
  memcpy(h2e, eri, sizeof(double) * n4);
  for (int k = 0; k < norb; ++k)
    for (int p = 0; p < norb; ++p)
      for (int q = 0; q < norb; ++q) {
        h2e[k * n3 + k * n2 + p * n1 + q] += f[p * n1 + q];
        h2e[p * n3 + q * n2 + k * n1 + k] += f[p * n1 + q];
      }
  for (int64_t i = 0; i < n4; ++i) h2e[i] *= fac;
  free(f);
}

/* same-spin part on a matrix X[nrow][ncol] whose ROWS carry the spin being excited:
 * for each intermediate k:  T[qs,:] = sum sg' X[t',:];  V = g T;  out[t,:] += sg V[pr,:]   (SCIcontract_2e_aaaa) */
static void contract_aaaa(const double* g, int nnorb_a, const double* X, double* out, int64_t nrow, int64_t ncol,
                          int64_t ninter, int nlink, const int32_t* dd) {
  (void)nrow;
#pragma omp parallel
  {
    double* T = (double*)malloc(sizeof(double) * nnorb_a * ncol);
    double* V = (double*)malloc(sizeof(double) * nnorb_a * ncol);
    double* acc = (double*)calloc((size_t)nrow * ncol, sizeof(double));
#pragma omp for schedule(dynamic, 4)
    for (int64_t k = 0; k < ninter; ++k) {
      const int32_t* tab = dd + k * nlink * 4;
      if (tab[3] == 0) continue;
      memset(T, 0, sizeof(double) * nnorb_a * ncol);
      for (int j = 0; j < nlink && tab[j * 4 + 3] != 0; ++j) {
        const double sg = tab[j * 4 + 3];
        const double* xr = X + (int64_t)tab[j * 4 + 2] * ncol;
        double* tr = T + (int64_t)tab[j * 4 + 0] * ncol;
        for (int64_t b = 0; b < ncol; ++b) tr[b] += sg * xr[b];
      }
      gemm_nn(nnorb_a, ncol, nnorb_a, g, T, V); /* dense, as pyscf does */
      for (int j = 0; j < nlink && tab[j * 4 + 3] != 0; ++j) {
        const double sg = tab[j * 4 + 3];
        double* orow = acc + (int64_t)tab[j * 4 + 2] * ncol;
        const double* vr = V + (int64_t)tab[j * 4 + 0] * ncol;
        for (int64_t b = 0; b < ncol; ++b) orow[b] += sg * vr[b];
      }
    }
#pragma omp critical
    for (int64_t i = 0; i < nrow * ncol; ++i) out[i] += acc[i];
    free(T); free(V); free(acc);
  }
}

/* A.4: sigma = H c.  h2e = ref_absorb_h1e(..., 0.5) as a full norb^4 array. */
void ref_contract_2e(double* sigma, const double* c, const double* h2e, int norb, int nelec_a, int nelec_b,
                     int64_t na, int64_t nb, int nlinka, const int32_t* cda, int nlinkb, const int32_t* cdb,
                     int64_t nintera, int ndla, const int32_t* dda, int64_t ninterb, int ndlb, const int32_t* ddb) {
  const int64_t n1 = norb, n2 = n1 * n1, n3 = n2 * n1;
  const int nnorb_a = norb * (norb - 1) / 2, nnorb_s = norb * (norb + 1) / 2;
  const int64_t D = na * nb;
  memset(sigma, 0, sizeof(double) * D);
  /* (i) per-call re-packing: g[(p>r),(q>s)] = 2 (h2e[pqrs] - h2e[psrq]) */
  double* g = (double*)malloc(sizeof(double) * (nnorb_a > 0 ? (int64_t)nnorb_a * nnorb_a : 1));
  for (int p = 1; p < norb; ++p)
    for (int r = 0; r < p; ++r)
      for (int q = 1; q < norb; ++q)
        for (int s = 0; s < q; ++s)
          g[(int64_t)(p * (p - 1) / 2 + r) * nnorb_a + (q * (q - 1) / 2 + s)] =
              2.0 * (h2e[p * n3 + q * n2 + r * n1 + s] - h2e[p * n3 + s * n2 + r * n1 + q]);
  /* (ii) beta-beta on C^T, alpha-alpha on C */
  if (nelec_b > 1 && ninterb > 0) {
    double* ct = (double*)malloc(sizeof(double) * D);
    double* st = (double*)calloc(D, sizeof(double));
    for (int64_t a = 0; a < na; ++a)
      for (int64_t b = 0; b < nb; ++b) ct[b * na + a] = c[a * nb + b];
    contract_aaaa(g, nnorb_a, ct, st, nb, na, ninterb, ndlb, ddb);
    for (int64_t a = 0; a < na; ++a)
      for (int64_t b = 0; b < nb; ++b) sigma[a * nb + b] += st[b * na + a];
    free(ct); free(st);
  }
  if (nelec_a > 1 && nintera > 0) contract_aaaa(g, nnorb_a, c, sigma, na, nb, nintera, ndla, dda);
  free(g);
  /* (iii) alpha-beta: e1 = 2 h2e, + hps/na on (..|kk), + hps/nb on (kk|..); packed tril x tril.
   * first pair index acts on beta, second on alpha */
  double* hps = (double*)calloc(n2, sizeof(double));
  for (int p = 0; p < norb; ++p)
    for (int s = 0; s < norb; ++s) {
      double v = 0.0;
      for (int q = 0; q < norb; ++q) v += h2e[p * n3 + q * n2 + q * n1 + s];
      hps[p * n1 + s] = v;
    }
  double* e1 = (double*)malloc(sizeof(double) * (int64_t)nnorb_s * nnorb_s);
  for (int p = 0; p < norb; ++p)
    for (int q = 0; q <= p; ++q)
      for (int r = 0; r < norb; ++r)
        for (int s = 0; s <= r; ++s) {
          double v = 2.0 * h2e[p * n3 + q * n2 + r * n1 + s];
          if (r == s) v += hps[p * n1 + q] / (nelec_a + 1e-100);
          if (p == q) v += hps[r * n1 + s] / (nelec_b + 1e-100);
          e1[(int64_t)(p * (p + 1) / 2 + q) * nnorb_s + (r * (r + 1) / 2 + s)] = v;
        }
  free(hps);
#pragma omp parallel
  {
    double* T = (double*)malloc(sizeof(double) * nnorb_s * nb); /* T[rs][b] */
    double* V = (double*)malloc(sizeof(double) * nnorb_s * nb); /* V[pq][b] */
#pragma omp for schedule(dynamic, 2)
    for (int64_t A = 0; A < na; ++A) {
      memset(T, 0, sizeof(double) * nnorb_s * nb);
      const int32_t* ta = cda + A * nlinka * 4;
      for (int j = 0; j < nlinka && ta[j * 4 + 3] != 0; ++j) {
        const double sg = ta[j * 4 + 3];
        const double* cr = c + (int64_t)ta[j * 4 + 2] * nb;
        double* tr = T + (int64_t)ta[j * 4 + 0] * nb;
        for (int64_t b = 0; b < nb; ++b) tr[b] += sg * cr[b];
      }
      gemm_nn(nnorb_s, nb, nnorb_s, e1, T, V); /* V[pq][b] = sum_rs e1[pq][rs] T[rs][b] */
      double* srow = sigma + A * nb;
      for (int64_t B = 0; B < nb; ++B) {
        const int32_t* tb = cdb + B * nlinkb * 4;
        double acc = 0.0;
        for (int j = 0; j < nlinkb && tb[j * 4 + 3] != 0; ++j)
          acc += tb[j * 4 + 3] * V[(int64_t)tb[j * 4 + 0] * nb + tb[j * 4 + 2]];
        srow[B] += acc;
      }
    }
    free(T); free(V);
  }
  free(e1);
}

/* A.7: occupancies = diag of rdm1s */
void ref_occupancies(double* occ_a, double* occ_b, const double* c, int norb, int64_t na, int64_t nb,
                     const uint64_t* sa, const uint64_t* sb) {
  for (int p = 0; p < norb; ++p) occ_a[p] = occ_b[p] = 0.0;
  for (int64_t a = 0; a < na; ++a)
    for (int64_t b = 0; b < nb; ++b) {
      const double w = c[a * nb + b] * c[a * nb + b];
      for (int p = 0; p < norb; ++p) {
        if ((sa[a] >> p) & 1ull) occ_a[p] += w;
        if ((sb[b] >> p) & 1ull) occ_b[p] += w;
      }
    }
}
