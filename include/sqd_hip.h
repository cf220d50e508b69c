/* sqd_hip.h -- C ABI of libsqd_hip.so: the MI355X (gfx950) implementation of the
 * fermionic subspace projection + diagonalization hot path of qiskit-addon-sqd.
 *
 * The reference reaches this arithmetic through pyscf's ctypes boundary into
 * libfci.so (reference qiskit_addon_sqd/fermion.py:721-729, :810-830, :117-133).
 * Each entry point below names the reference call (file:line) and the pyscf
 * routine it replaces.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions:
 *   - CI string: uint64, bit p = occupation of spatial orbital p (LSB = orbital 0).
 *   - basis = {|a>|b>}: a in strs_a (sorted, unique), b in strs_b (sorted, unique);
 *     amplitudes C[ia*nb + ib] (row = alpha string), float64.
 *   - h1[p*norb+q], eri[((p*norb+q)*norb+r)*norb+s] = (pq|rs) chemist order, 8-fold symmetric.
 *   - every function returns 0 on success, <0 on error; sqd_last_error() gives the text
 *     (thread-local).  Host buffers are caller-owned and never written unless documented
 *     as outputs.  One context = one device + one HIP stream; contexts are independent,
 *     so one host thread per context/device is safe.  A single context is not re-entrant.
 */
#ifndef SQD_HIP_H
#define SQD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sqd_ctx sqd_ctx;

#define SQD_OK 0
#define SQD_ERR_INVALID (-1)  /* bad argument / inconsistent Hamming weight / unsorted strings */
#define SQD_ERR_HIP (-2)      /* HIP runtime failure */
#define SQD_ERR_STATE (-3)    /* call sequence error (e.g. no subspace set) */
#define SQD_ERR_LIMIT (-4)    /* size beyond what this build supports */

#define SQD_MAX_NORB 64
#define SQD_MAX_SPACE 30

/* ABI version of this header (bumped on any signature change). */
int sqd_abi_version(void);
const char* sqd_last_error(void);
int sqd_device_count(int* count);

/* Page-locked host memory for result buffers (hipHostMalloc / hipHostFree).  An `amps` buffer obtained here is
 * written by the GPU directly (sqd_solve, sqd_solve_strings: by the observables kernel up to 64 MB, by the DMA engine
 * beyond); any other pointer goes through an internal pinned staging buffer and a host copy.  No counterpart in the reference (pyscf returns numpy arrays it allocates itself). */
int sqd_host_alloc(size_t bytes, void** out);
int sqd_host_free(void* p);

/* Create a solver context on `device` holding the Hamiltonian integrals.
 * Replaces: pyscf SelectedCI() construction + direct_spin1.absorb_h1e + ao2mo.restore
 * done inside kernel_fixed_space (reference fermion.py:713,721 / :803,810). */
int sqd_ctx_create(int device, int norb, const double* h1, const double* eri, sqd_ctx** out);
int sqd_ctx_destroy(sqd_ctx* ctx);
/* Make the context enqueue all its work on a caller-owned HIP stream (a hipStream_t passed as void*, e.g.
 * torch.cuda.Stream().cuda_stream) instead of the stream it created.  For callers with device work of
 * their own per solve -- the RCCL exchange of the per-batch records [E, occ_a, occ_b] after the
 * reference's batch loop (fermion.py:432, :577-605): on the solver's stream the exchange costs 36 us per
 * step, on another hardware queue 77-110 us (profiles/probes/_exchange_probe.py).  The stream must
 * outlive the context; call between solves (the previous stream is drained first).  No counterpart in
 * the reference: pyscf is host code. */
int sqd_ctx_use_stream(sqd_ctx* ctx, void* stream);
/* From now on every sqd_solve / sqd_solve_strings / sqd_solve_batch of this context ALSO leaves the raw record of its
 * observables in device memory at d_record (batch p of sqd_solve_batch: d_record + p * stride doubles):
 *   {e_davidson, c.Hc (0 here), c.S2c, c.c, occ_a[norb], occ_b[norb], |S2 c|^2}  -- 5 + 2 norb doubles, un-normalised;
 * energy = e_davidson - shift * penalty, occupancies = occ / c.c exactly as the call's host outputs are formed.
 * Written by the observables kernel itself, so a collective on the same stream (sqd_ctx_use_stream) -- the RCCL
 * all-reduce that makes every rank know every batch's record after the reference's collective step, fermion.py:432,
 * :577-605 -- needs no host-to-device copy and no host wait in between.  NULL switches it off (the default). */
int sqd_ctx_set_record_out(sqd_ctx* ctx, double* d_record, int64_t stride);
/* A function the solve calls of this context (sqd_solve / sqd_solve_strings / sqd_solve_batch) call ONCE per call, on the
 * calling thread, when their last kernel has been enqueued and before they wait for it: whatever the hook enqueues on the
 * context's stream -- the collective that follows the reference's batch loop (fermion.py:432), the copy of its result --
 * runs right behind the solve instead of one host wait and one enqueue later (the exchange of the N > 1 path: 15-20 us
 * of idle stream per step).  The hook must not call into this context.  NULL removes it (the default). */
typedef void (*sqd_enqueue_hook)(void* user);
int sqd_ctx_set_enqueue_hook(sqd_ctx* ctx, sqd_enqueue_hook hook, void* user);

/* Define the subspace.  strs_a / strs_b must be strictly ascending with a constant
 * popcount per spin (the post-condition of reference _check_ci_strs, fermion.py:1075-1097);
 * violations return SQD_ERR_INVALID.  Builds on the device: the single-/double-excitation
 * link tables with bit-exact addresses and signs (pyscf _all_linkstr_index: SCIcre_des_linkstr,
 * SCIdes_des_linkstr), the diagonal (pyscf make_hdiag / FCImake_hdiag_uhf) and the per-string
 * mean-field tables used by the sigma kernel.  nelec is taken from the popcounts. */
int sqd_set_subspace(sqd_ctx* ctx, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb);

/* ---- intra-solve sharding by alpha rows (the collective sci_solver contract of reference
 * docs/guides/hpc_acceleration.rst:52-57).  One process per GPU; rank r calls sqd_set_subspace_rows with ITS row range
 * [row0, row1) of the same (strs_a, strs_b): link tables cover all strings, hdiag and the sigma work list only the
 * owned rows.  sqd_sigma_rows_dev / sqd_contract_ss_rows_dev then map the FULL vector c (na*nb doubles, gathered by
 * the caller -- RCCL all-gather over xGMI in qiskit_addon_sqd_amd.sharded) to rows [row0, row1) of sigma / S^2 c
 * ((row1-row0)*nb doubles).  These three take DEVICE pointers (memory of the context's device) and only enqueue work on
 * the context's stream; sqd_ctx_sync waits for it.  Whole-vector entry points (sqd_sigma, sqd_davidson, sqd_solve,
 * the observables) return SQD_ERR_STATE on a sharded context; sqd_hdiag returns the owned rows. */
int sqd_set_subspace_rows(sqd_ctx* ctx, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb,
                          int64_t row0, int64_t row1);
int sqd_sigma_rows_dev(sqd_ctx* ctx, const double* d_c_full, double* d_sigma_rows, int use_spin, double ss, double shift);
int sqd_contract_ss_rows_dev(sqd_ctx* ctx, const double* d_c_full, double* d_out_rows);
int sqd_hdiag_rows_dev(sqd_ctx* ctx, double* d_out_rows);
int sqd_ctx_sync(sqd_ctx* ctx);
/* Device address of the resident Davidson solution (na*nb doubles, normalised), valid until the next solve or
 * sqd_set_subspace on this context.  For collectives that ship the winning state between GPUs without a detour through
 * the host (the broadcast after the reference's batch loop, fermion.py:432 / :608-631). */
int sqd_solution_device_ptr(sqd_ctx* ctx, const double** d_ptr);
/* Copy the resident solution of the latest sqd_davidson / sqd_solve (called with amps == NULL) to `amps`
 * (na*nb doubles): the state on demand, for callers that leave it on the device by default. */
int sqd_solution_copy(sqd_ctx* ctx, double* amps);
/* on != 0: sqd_solve / sqd_solve_strings return as soon as the RESULTS (energy, <S^2>, occupancies, statistics) are on the
 * host; the amplitudes follow into `amps` -- which must then come from sqd_host_alloc and be at most 64 MB, else the call
 * behaves as before -- written by a kernel of their own on the context's copy stream, started behind the solution (0.8 MB
 * at the headline size are 24 us of posted PCIe writes: longer than every other kernel of the solve, and now beside the
 * next solve's table build instead of in front of it).  stats->state_ticket > 0 says so;
 * sqd_ctx_state_wait(ctx, ticket) returns when the buffer is complete (a later ticket of the same context implies every
 * earlier one).  The reference returns the numpy array with the call (fermion.py:724, :820); the Python layer wraps the
 * pending buffer so that the first READ of SCIState.amplitudes waits.  Default: off. */
int sqd_ctx_set_async_state(sqd_ctx* ctx, int on);
int sqd_ctx_state_wait(sqd_ctx* ctx, long long ticket);
/* on != 0: bracket every following sqd_set_subspace and Davidson run of this context with HIP events, so that
 * sqd_davidson_stats::ms_setup / ms_total are filled.  Off by default: each event record is a bubble in a stream of
 * ~5 us kernels (four records cost ~40 us of a 0.2 ms solve).  The sigma-launch sampling of time_sigma_every is
 * independent of this switch. */
int sqd_ctx_set_phase_timing(sqd_ctx* ctx, int on);

/* Sizes of the current subspace. */
int sqd_get_dims(sqd_ctx* ctx, int64_t* na, int64_t* nb, int* nelec_a, int* nelec_b);

/* Link tables, for bit-exact addressing tests.  spin: 0 = alpha, 1 = beta.
 * Singles: |strs[tgt]> = sign * a+_cre a_des |strs[src]>, sorted by (tgt, src);
 *          pair = tril index max(max+1)/2+min of (cre,des) (pyscf cre_des_linkstr_tril column 0).
 * Doubles: |strs[tgt]> = sign * a+_p a+_r a_s a_q |strs[src]>, p>r, q>s, sorted by (tgt, src);
 *          orbs[4*i..] = p, r, q, s.
 * Pass NULL output pointers to query counts only. */
int sqd_link_counts(sqd_ctx* ctx, int spin, int64_t* n_single, int64_t* n_double);
int sqd_single_links(sqd_ctx* ctx, int spin, int32_t* tgt, int32_t* src, int32_t* cre, int32_t* des,
                     int32_t* pair, int32_t* sign, double* value);
int sqd_double_links(sqd_ctx* ctx, int spin, int32_t* tgt, int32_t* src, int32_t* orbs, int32_t* sign,
                     double* value);

/* hdiag[ia*nb+ib] = <ab|H|ab>.  Replaces pyscf SelectedCI.make_hdiag (kernel_fixed_space). */
int sqd_hdiag(sqd_ctx* ctx, double* out);

/* Davidson start vector used when ci0 == NULL: pyscf SelectedCI.get_init_guess -> direct_spin1._get_init_guess
 * (inside kernel_fixed_space, reference fermion.py:721, :810): unit vector at the lowest diagonal element --
 * searched over the lower triangle ia >= ib when nelec_a == nelec_b and na == nb -- with +1e-5 on the first and
 * -1e-5 on the last element, normalised.  out: na*nb doubles. */
int sqd_init_guess(sqd_ctx* ctx, double* out);

/* sigma = P H P c  (+ penalty).  Replaces pyscf selected_ci.contract_2e
 * (SCIcontract_2e_aaaa x2, SCIcontract_2e_bbaa) and, when use_spin != 0, the fix_spin_ wrapper:
 *   use_spin = 0 : sigma = H c
 *   use_spin = 1 : sigma = H c + shift * (S^2 - ss) c            (pyscf form for ss < sz(sz+1)+0.1)
 *   use_spin = 2 : sigma = H c + shift * (S^2 - ss)^2 c          (pyscf form otherwise)
 *   use_spin = 3 : let the library choose 1 or 2 by pyscf's rule. */
int sqd_sigma(sqd_ctx* ctx, const double* c, double* sigma, int use_spin, double ss, double shift);

/* out = P S^2 P c.  Replaces pyscf selected_ci.contract_ss. */
int sqd_contract_ss(sqd_ctx* ctx, const double* c, double* out);

typedef struct sqd_davidson_opts {
  double tol;        /* pyscf conv_tol, default 1e-9 (SelectedCI) */
  double tol_residual; /* |r| threshold; <= 0 selects pyscf's rule sqrt(tol) when use_spin == 0 (<c|H|c> is second
                          order in the residual: the reference's own accuracy) and sqrt(tol)/32 with a spin penalty
                          (the returned <c|H|c> = Ritz value - shift <penalty> is FIRST order in it).  Orbital
                          occupancies are first order in the residual either way: pass sqrt(tol)/32 to have them at
                          1e-6 instead of 1e-4 (about a third more sigma builds). */
  double lindep;     /* 1e-14 */
  int max_cycle;     /* 100 */
  int max_space;     /* 12 */
  int use_spin;      /* 0 none, 3 = pyscf fix_spin_ rule */
  double ss;         /* target S^2 value */
  double shift;      /* penalty strength (0.1 solve_fermion, 0.2 solve_sci) */
  int verbose;
  int time_sigma_every; /* k > 0: bracket every k-th sigma launch of this context with HIP events (fills
                           ms_sigma / n_sigma_timed; an event pair costs ~10 us of stream time); 0: none */
} sqd_davidson_opts;

typedef struct sqd_davidson_stats {
  int converged;
  int iterations;
  int n_sigma;         /* number of sigma builds */
  double e_davidson;   /* eigenvalue of H + penalty (pyscf's discarded return value, without ecore) */
  double residual;     /* final |r| */
  double ms_total;     /* device time of the Davidson loop (HIP events on the context stream) */
  double ms_sigma;     /* device time summed over the TIMED sigma applications (k_sigma + k_sigma_reduce;
                          see time_sigma_every) */
  double ms_setup;     /* device time of the last sqd_set_subspace */
  int n_sigma_timed;   /* number of sigma launches bracketed by events in this run */
  double ms_sigma_kernel; /* of which: the k_sigma launches alone (start event .. event after k_sigma) */
  double ms_event_overhead; /* summed duration of the EMPTY event bracket recorded right behind every timed sigma: what
                               two event records cost by themselves; ms_sigma_kernel minus this is the kernel time */
  int n_eig_solves;    /* device-side projected eigenproblem: shifted solves of the warm-started Rayleigh-quotient iteration */
  int n_eig_fallbacks; /* ... and how many projected problems fell back to the Jacobi solver */
  long long state_ticket; /* sqd_solve / sqd_solve_strings on a context with sqd_ctx_set_async_state(ctx, 1): > 0 when the
                             amplitudes were still on their way to `amps` at return -- sqd_ctx_state_wait(ctx, ticket) */
} sqd_davidson_stats;

void sqd_davidson_default_opts(sqd_davidson_opts* o);

/* Ground state of P (H + penalty) P by Davidson, resident on the device.
 * Replaces pyscf kernel_fixed_space -> FCISolver.eig -> lib.davidson1 (reference
 * fermion.py:721-723, :810-818).  ci0 (na*nb doubles) may be NULL: pyscf get_init_guess is used.
 * The normalised solution stays resident in the context (used by the observables below) and is
 * also copied to `amps` when it is not NULL. */
int sqd_davidson(sqd_ctx* ctx, const sqd_davidson_opts* opts, const double* ci0, double* amps,
                 sqd_davidson_stats* stats);

/* The Davidson of a row-sharded subspace, stage by stage.  Every rank runs the SAME device-resident state machine as
 * sqd_davidson (projected matrix, lowest eigenpair, restart, stop rule: pyscf lib.davidson1 as called at reference
 * fermion.py:721-723) on its rows; the reductions are cut where a number crosses ranks -- each stage that needs one
 * leaves its LOCAL totals in a small device buffer and returns its address and length, the caller all-reduces (sum) that
 * buffer on the context's stream (sqd_ctx_use_stream; RCCL in qiskit_addon_sqd_amd.sharded) and calls the next stage.
 * One iteration:  pick (rows of the newest basis vector -> send buffer; the caller all-gathers it into the full vector)
 * -> sigma (rows of H c from the full vector) -> dots -> [all-reduce] -> residual (eigenproblem, residual,
 * preconditioner) -> [all-reduce] -> orth (stop rule, Gram-Schmidt, restart collapse; returns a ticket).  No stage waits
 * for the device: every kernel reads "which vector / whether to stop" from the state block, so the caller may enqueue
 * iteration k + 1 before sqd_shard_dav_wait(ticket of k) -- the one host wait -- tells it whether the run has stopped
 * (stages enqueued behind a stop return at once).  begin returns where the rows of the (normalised) start vector go; end returns where
 * the rows of the solution are.  The squared spin penalty (use_spin = 2) is refused: it chains S^2 through a second
 * gathered vector per sigma build (qiskit_addon_sqd_amd.sharded keeps a torch-level driver for it). */
int sqd_shard_dav_begin(sqd_ctx* ctx, const sqd_davidson_opts* opts, double** d_x0_rows);
int sqd_shard_dav_pick(sqd_ctx* ctx, double** d_send_rows);
int sqd_shard_dav_sigma(sqd_ctx* ctx, const double* d_c_full);
/* The sigma stage in two calls around the all-gather, so that the gather overlaps the work that needs no remote row
   (reference contract: docs/guides/hpc_acceleration.rst:52-57, the collective sci_solver): part 1 -- d_c_full may be
   NULL -- right behind sqd_shard_dav_pick, on the rows in the send buffer; part 2 with the gathered vector.  part 0 =
   sqd_shard_dav_sigma.  Bit-identical to the one-call stage. */
int sqd_shard_dav_sigma_part(sqd_ctx* ctx, const double* d_c_full, int part);
int sqd_shard_dav_dots(sqd_ctx* ctx, double** d_totals, int* count);
int sqd_shard_dav_residual(sqd_ctx* ctx, double** d_totals, int* count);
int sqd_shard_dav_orth(sqd_ctx* ctx, long long* ticket);
/* ... and pick + sigma + dots + eigen step + residual + orth as ONE call, for a context that holds all rows (a group of one
 * rank: every collective between the stages is the identity); SQD_ERR_STATE on a true shard.  Same bits as the stages. */
int sqd_shard_dav_iteration(sqd_ctx* ctx, long long* ticket);
int sqd_shard_dav_wait(sqd_ctx* ctx, long long ticket, int* stopped, double* e, double* rnorm2, int* basis_size);
int sqd_shard_dav_end(sqd_ctx* ctx, double** d_solution_rows, sqd_davidson_stats* stats);

/* Observables of a state.  amps == NULL means "the resident Davidson solution".
 * sqd_energy: <c|H|c> without penalty (what the reference recomputes from RDMs, fermion.py:730-732,:827).
 * sqd_spin_square: <c|S^2|c> (pyscf spin_square, fermion.py:830,:133).
 * sqd_rdm1s: dm1a/dm1b[p*norb+q] = <a+_p a_q> per spin (pyscf make_rdm1s, fermion.py:725,:821,:121).
 * sqd_rdm2: spin-summed dm2[p,q,r,s] = sum_{st} <p+_s r+_t s_t q_s> (pyscf make_rdm2, fermion.py:729,:826). */
/* sqd_observables: everything reference solve_fermion derives from the state after the Davidson
 * (fermion.py:820-830) in one call with one host synchronisation: e = <c|H|c>/<c|c>, s2 = <c|S^2|c>/<c|c>,
 * occ_a/occ_b[norb] = diagonals of dm1a/dm1b. */
int sqd_observables(sqd_ctx* ctx, const double* amps, double* e, double* s2, double* occ_a, double* occ_b);

/* sqd_solve = sqd_davidson + sqd_observables of the solution, as one call: everything reference
 * solve_fermion does between building the solver and returning (fermion.py:803-830).  One fused observables kernel;
 * it also writes the amplitudes into `amps` when that buffer came from sqd_host_alloc (else a copy on a second stream
 * overlaps it); one host wait.
 * Any of amps / stats / e / s2 / occ_a / occ_b may be NULL. */
int sqd_solve(sqd_ctx* ctx, const sqd_davidson_opts* opts, const double* ci0, double* amps,
              sqd_davidson_stats* stats, double* e, double* s2, double* occ_a, double* occ_b);
/* sqd_solve_strings = sqd_set_subspace + sqd_solve in one call (one crossing of the ctypes boundary per solve): what
 * reference solve_fermion does from the checked CI strings to its return value (fermion.py:797-830) and solve_sci
 * per batch (fermion.py:713-742).  amps: na*nb doubles.  nelec_a / nelec_b (may be NULL) receive the popcounts. */
int sqd_solve_strings(sqd_ctx* ctx, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb,
                      const sqd_davidson_opts* opts, const double* ci0, double* amps, sqd_davidson_stats* stats,
                      double* e, double* s2, double* occ_a, double* occ_b, int* nelec_a, int* nelec_b);

/* ---- the whole list of subspaces the reference hands its sci_solver at once (fermion.py:432; solved one after
 * another by reference solve_sci_batch, fermion.py:670-681, "embarrassingly parallel" README.md:76) as ONE batched
 * solve: the table build, every Davidson round and the observables of ALL nbatch subspaces advance in the same set of
 * kernel launches (blockIdx.z = subspace) on the context's stream; each subspace keeps its own device-resident Davidson
 * state and stops itself, the call returns when all have.  Per subspace the arithmetic is sqd_solve_strings' -- same
 * kernels bodies on the same per-subspace grids -- so every output equals the one-by-one result bit for bit.
 *   strs_a[i] / na[i] / strs_b[i] / nb[i]: the strings of batch i (as sqd_set_subspace).
 *   e / s2 / nelec_a / nelec_b: nbatch values; occ_a / occ_b: nbatch * norb doubles, row i = batch i; stats: nbatch
 *   records.  s2 may be NULL (then <S^2> is not evaluated unless the spin penalty needs it); so may stats, nelec_*.
 *   amps: NULL, or nbatch pointers; amps[i] (NULL = leave the state on the device) receives batch i's na[i]*nb[i]
 *   amplitudes -- written by the observables kernel itself when it came from sqd_host_alloc.
 *   best_amps (may be NULL): room for the largest subspace; receives the amplitudes of the lowest-energy batch --
 *   the only state the reference's loop consumes (fermion.py:577, :608-631) -- and *best its index.  Every other
 *   state stays resident for this and the NEXT sqd_solve_batch on this context: sqd_batch_state copies one out on
 *   demand.
 * opts->verbose and opts->time_sigma_every are ignored; ci0 is not available (pyscf's start vector is used).
 * Subspaces outside the batched launch classes (rows too long for LDS staging, the (S^2-ss)^2 penalty form) are
 * solved one by one inside the same call. */
int sqd_solve_batch(sqd_ctx* ctx, int nbatch, const uint64_t* const* strs_a, const int64_t* na,
                    const uint64_t* const* strs_b, const int64_t* nb, const sqd_davidson_opts* opts,
                    double* const* amps, double* best_amps, int* best, sqd_davidson_stats* stats, double* e, double* s2,
                    double* occ_a, double* occ_b, int* nelec_a, int* nelec_b);
/* amplitudes of batch `index` (na*nb doubles) from their device-resident copy: age 0 = of the latest sqd_solve_batch,
 * age 1 = of the one before it (each subspace slot keeps two solutions, so the results of call N stay readable while
 * and after call N + 1 runs -- the `results = solver(...)` loop of the reference, fermion.py:432) */
int sqd_batch_state(sqd_ctx* ctx, int index, int age, double* amps);
/* the sub-context that holds batch `index` of the latest sqd_solve_batch (tables + resident solution): usable with the
 * observables / RDM entry points (amps = NULL: the resident solution) until the next sqd_solve_batch.  Owned by ctx. */
int sqd_batch_ctx(sqd_ctx* ctx, int index, sqd_ctx** sub);
int sqd_energy(sqd_ctx* ctx, const double* amps, double* e);
int sqd_spin_square(sqd_ctx* ctx, const double* amps, double* s2);
int sqd_rdm1s(sqd_ctx* ctx, const double* amps, double* dm1a, double* dm1b);
int sqd_rdm2(sqd_ctx* ctx, const double* amps, double* dm2);
/* sqd_rdm2s: the spin-resolved pieces (dm2aa, dm2ab, dm2bb), dm2ab[p,q,r,s] = <p+_a r+_b s_b q_a> (pyscf
 * make_rdm2s, reference fermion.py:124-125 -- SCIState.rdm(rank=2, spin_summed=False)); norb^4 doubles each.
 * dm2 of sqd_rdm2 = dm2aa + dm2bb + dm2ab + dm2ab^T(2,3,0,1). */
int sqd_rdm2s(sqd_ctx* ctx, const double* amps, double* dm2aa, double* dm2ab, double* dm2bb);

/* Benchmark hooks: run `reps` sigma builds on the resident solution buffer and report the
 * average device time per launch of the dominant sigma kernel (HIP events on the context stream). */
int sqd_time_sigma(sqd_ctx* ctx, int reps, int use_spin, double ss, double shift, double* ms_per_sigma);
/* ... with a HIP-event bracket around every launch of the dominant sigma kernel and an empty bracket behind it:
 * out4 = {mean, median kernel bracket, mean, median empty bracket} in ms over `reps` (<= 4096) launches. */
int sqd_time_sigma_brackets(sqd_ctx* ctx, int reps, int use_spin, double ss, double shift, double* out4);
/* ... and of the matrix-core same-spin product alone (subspaces in dense mode, sqd_sigma_kernel kind 3): `copies` identical
 * problems per launch (1 = what a single solve launches, 16 = a batched solve's full chip), average device time per
 * launch and the flops of one launch on the padded orders (2 pa^2 pb + 2 pa pb^2 per copy) -- the MFMA roofline entry of
 * bench.py. */
int sqd_time_dense(sqd_ctx* ctx, int reps, int copies, double* ms_per_launch, double* flops_per_launch);
/* Populated-link count and algorithmic bytes of one sigma (SURVEY 8d formula) for the current subspace. */
int sqd_sigma_bytes(sqd_ctx* ctx, double* bytes);
/* Bytes this build's formulation has to move once per sigma (vectors + hdiag + the link records at their stored
 * width + the integral / J rows the populated links touch): the honest denominator beside SURVEY 8d's B_sigma, which
 * charges the whole packed integral tables whether or not a string set has the links that read them. */
int sqd_sigma_bytes_needed(sqd_ctx* ctx, double* bytes);
/* Which sigma kernel the current subspace selected (benchmark / test hook, no reference counterpart):
 * kind 0 = work items (k_sigma), 1 = element gather (k_sigma_direct), 2 = whole rows in LDS (k_sigma_rows),
 * 3 = dense same-spin blocks on the f64 matrix cores (k_same_spin_mfma) + work items for the opposite-spin terms,
 * 4 = list passes for large sets with short lists (k_sigma_lists: link lists in registers, rows of C and C^T through LDS),
 * 5 = sparse same-spin product (k_spmm_*) + work items, 6 = sparse product + whole-row opposite-spin kernel (k_opp_rows,
 * rows of <= 3072 columns), 7 = sparse product + the source-range form of it (k_opp_src, longer rows);
 * rows_per_workgroup: kind 2 -> rows of C per workgroup; kinds 5-7 -> rows per group of the sparse product (8 =
 * k_spmm_grouped, 1 = k_spmm_rows); else 0. */
int sqd_sigma_kernel(sqd_ctx* ctx, int* kind, int* rows_per_workgroup);

/* ---- qubit / Pauli path (reference qiskit_addon_sqd/qubit.py) -----------------
 * Projection of sum_t c_t P_t onto the subspace spanned by the computational basis states `rows`
 * (strictly ascending unsigned integers of the bitstrings, column 0 = most significant bit; d of them).
 * Terms are grouped by x mask: group g has mask xmask[g] and the terms group_ptr[g] .. group_ptr[g+1];
 * term t has z mask zmask[t] and coefficient coef[2t] + i coef[2t+1] which must already include
 * i^{popcount(x & z)} (the Y phase).  Result is CSR with row = input configuration and column =
 * connected configuration, A[r, index(rows[r] ^ x)] = sum_t coef_t (-1)^{popcount(rows[r] & z_t)} -- the
 * summed output of reference matrix_elements_from_pauli (qubit.py:167-240) over the term loop of
 * project_operator_to_subspace (qubit.py:127-142).
 * sqd_pauli_count: uploads, counts, scans; returns indptr[d+1] (may be NULL), nnz and a plan.
 * sqd_pauli_fill : indices[nnz], data[2*nnz] (re, im interleaved); ms_kernels = device time of both passes.
 * sqd_pauli_free : releases the plan. */
typedef struct sqd_pauli_plan sqd_pauli_plan;
int sqd_pauli_count(int device, const uint64_t* rows, int64_t d, int ngroups, const uint64_t* xmask,
                    const int64_t* group_ptr, const uint64_t* zmask, const double* coef, int64_t* indptr_out,
                    int64_t* nnz_out, sqd_pauli_plan** plan_out);
int sqd_pauli_fill(sqd_pauli_plan* plan, int64_t* indices, double* data, double* ms_kernels);
int sqd_pauli_free(sqd_pauli_plan* plan);

/* ---- host-side half of configuration recovery (no device work) -------------------------------------------
 * Replaces the per-bitstring Python loop of reference configuration_recovery.py:230-304 (_bipartite_bitstring_
 * correcting): every listed row of the bool sample matrix `bits` ([n_total][2*norb], left half = spin-down) is
 * brought to Hamming weights (target_left, target_right) by flipping bits drawn without replacement with the
 * occupancy-informed weights.  The draws REPLAY numpy's Generator.choice(candidates, size, replace=False, p=p)
 * on a caller-supplied stream of uniform doubles (`uniforms`, from the same Generator), left half then right
 * half, row after row; *n_used = doubles consumed, so that the caller can rewind the generator by the rest
 * and a seeded run stays bit-identical to the reference's.  Returns SQD_ERR_STATE when a row's weights are
 * ones numpy would raise on (the caller then takes the slow path to raise the same exception), SQD_ERR_LIMIT
 * when the stream is too short. */
int sqd_recover_rows(uint8_t* bits, int64_t n_total, int norb, const int64_t* rows, int64_t nrows,
                     const double* up_left, const double* down_left, const double* up_right,
                     const double* down_right, int target_left, int target_right, const double* uniforms,
                     int64_t n_uniforms, int64_t* n_used);

/* The numpy passes around that repair, natively (no device work; same results, same stream position):
 * sqd_hamming_excess -- distance of both halves of every row from the target weights: *bound = upper bound on the doubles
 *   sqd_recover_rows (rows == NULL: all rows in order) can consume, *nbad = rows off target
 *   (configuration_recovery.py:231-241 computes the same sums row by row);
 * sqd_merge_rows -- duplicates merged in first-occurrence order, probabilities added in row order (the running dictionary
 *   of configuration_recovery.py:112-126): first[k] = row of the k-th distinct bitstring, freq[k] its summed probability;
 *   compact != 0 also moves the distinct rows to the front of `bits`, in that order;
 * sqd_choice_replay -- the `nbatches` calls Generator.choice(n, size, replace=False, p=p) of subsampling.py:200-207
 *   replayed one after the other on a block of uniforms, out[nbatches][size] (SQD_ERR_STATE: an input numpy raises on,
 *   SQD_ERR_LIMIT: stream too short). */
int sqd_hamming_excess(const uint8_t* bits, int64_t n, int norb, int target_left, int target_right, int64_t* bound,
                       int64_t* nbad);
int sqd_merge_rows(uint8_t* bits, int64_t n, int nbits, const double* probs, int64_t* first, double* freq,
                   int64_t* n_unique, int compact);
/* np.unique(bool_matrix, axis=0, return_counts=True) as counts.py:45-61 applies it to the shots of a BitArray: first[k] =
 * smallest row index of the k-th distinct row in lexicographic row order, counts[k] its multiplicity (host code; nbits <=
 * 128, else SQD_ERR_LIMIT; bytes other than 0 / 1: SQD_ERR_STATE). */
int sqd_unique_rows(const uint8_t* bits, int64_t n, int nbits, int64_t* first, int64_t* counts, int64_t* n_unique);
int sqd_choice_replay(const double* p, int64_t n, int64_t size, int64_t nbatches, const double* uniforms,
                      int64_t n_uniforms, int64_t* out, int64_t* n_used);

/* Digests of the caller's integral tensors on native threads (no device work).  The reference's entry points take the
 * tensors as plain arrays on every call (fermion.py:745-755) and never copy them; a solver context here is keyed by a hash
 * of every byte, so that an in-place edit between two calls is seen.  sqd_hash_start returns at once (a small pool of
 * native threads hashes fixed 512 KB pieces), sqd_hash_finish waits, helping, and returns one 64-bit digest per range
 * (p1 == NULL, n1 == 0: one range) and releases the job.  The ranges must stay alive and unchanged in between; jobs of
 * several host threads may be in flight at once.  The digest does not depend on the number of threads. */
int sqd_hash_start(const void* p0, size_t n0, const void* p1, size_t n1, void** job);
int sqd_hash_finish(void* job, unsigned long long* d0, unsigned long long* d1);

/* _check_ci_strs (fermion.py:1075-1097) without a dozen numpy passes for the lists every SQD iteration produces: *ok = 1
 * iff both lists are strictly ascending (= what np.sort(np.unique(.)) returns), non-negative as int64 and of one Hamming
 * weight each; *ok = 0 sends the caller to the numpy path, which raises the reference's errors or normalises. */
int sqd_check_strings(const uint64_t* a, int64_t na, const uint64_t* b, int64_t nb, int* ok);

#ifdef __cplusplus
}
#endif
#endif
