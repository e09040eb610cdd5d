/* libmagma_hip.so -- C ABI of the MI355X (gfx950, FP64) implementation of MAGMA's multi-task-GP
 * EM hot path. This is the drop-in boundary: what an R `.Call` shim, the
 * Python ctypes host in `magma_amd/`, or any other FFI binds. Plain C types only.
 *
 * Reference = ArthurLeroy/MAGMA: CF = Computing_functions.R, MG = MAGMA.R. The reference has no
 * FFI of its own (pure R closures); each entry point cites the R function whose body it
 * replaces.
 *
 * Conventions
 *  - All array arguments are DEVICE pointers unless named `h_*`. `stream` is a hipStream_t
 *    passed as void* (NULL = default stream). Calls enqueue work and return; results are valid
 *    after magma_stream_sync(stream) (or any later work on the same stream).
 *  - Return value: 0 = ok, >0 = error; magma_last_error() gives the message (thread-local).
 *    A covariance that is not numerically positive definite is NOT an error: the per-matrix
 *    `info` output is non-zero (index of the first bad pivot, LAPACK dpotrf style). Every SMALL
 *    (per-individual, n <= 64) matrix -- the E-step's K_i, magma_kern_to_inv_batched, the M-step
 *    objectives (magma_indiv_eval, magma_indiv_mstep, magma_train_new_gp) and cov_tn_tn of the
 *    predictions -- then goes through the reference's `tryCatch(solve(.), error -> MASS::ginv(.))`
 *    (CF:133,161,177,285; MG:173) on the device: LU with partial pivoting, and the pseudo-inverse
 *    (symmetric Jacobi, MASS::ginv's tolerance) when that is singular to machine precision --
 *    parity with R unpinned, no reference fixture reaches it. (The device-resident M-step reports
 *    status 6 for a problem whose start is flagged; see magma_indiv_mstep_resident.)
 *    DELIBERATE DEVIATION for the dense N x N objects (K_0, the hyper-posterior precision of
 *    CF:387 / MG:123, an individual with n > 64): no pseudo-inverse is computed. `info` reports
 *    the flag and the driver calls (magma_e_step, magma_em_iteration, magma_training, ...) return
 *    status 5 with a message naming the matrix. Reason: the precision K_0^-1 + sum_i K_i^-1 is a
 *    sum of positive definite matrices; it fails the sweep only when K_0 itself is singular to
 *    rounding (theta_0 at an extreme the optimiser should not be left at), where ginv() of an
 *    N = 2,000 .. 8,000 matrix would hand the M-step a rank-deficient cov that R itself then
 *    inverts again through the same tryCatch -- a result dominated by the 1.5e-8 cut-off, not by
 *    the data. The caller is told instead of being given that.
 *  - Ragged dataset: individual i owns rows off[i]..off[i+1]-1 of t / y / idx (timestamps
 *    ascending within an individual, as the reference assumes: CF:74,95). idx[] is the int32
 *    position of each timestamp in the sorted union grid (replaces the `paste0('X', t)` names of
 *    CF:78,451,474). n_i <= 64.
 *  - theta: per-individual (theta1, theta2, sigma) triples; theta_stride = 3, or 0 when every
 *    individual shares one triple (the "common HP" case, CF:89-93).
 *  - Dense matrices are Np x Np column-major with ld = Np = magma_pad(N) (N rounded up to a
 *    multiple of 64); rows/cols >= N hold the identity. Vectors on the grid have length Np.
 *  - One caller thread per stream; buffers must not be shared by concurrent calls.
 */
#ifndef MAGMA_HIP_H
#define MAGMA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library / device plumbing (no counterpart in the reference) ---- */
int magma_hip_version(void);
const char* magma_last_error(void);
int magma_pad(int N);
size_t magma_workspace_bytes(int N); /* scratch needed by the dense entry points below */
int magma_device_count(int* h_count);
int magma_set_device(int device);
int magma_dev_malloc(void** h_ptr, size_t bytes);
int magma_dev_free(void* ptr);
int magma_memcpy_h2d(void* dst, const void* h_src, size_t bytes, void* stream);
int magma_memcpy_d2h(void* h_dst, const void* src, size_t bytes, void* stream);
int magma_memset_zero(void* dst, size_t bytes, void* stream);
int magma_stream_sync(void* stream);

/* Optional timing of kernel classes with HIP events recorded on the launching stream
 * (bench.py's roofline figures). ids: 0 T = Kinv C GEMM, 1 theta_0 gradient contraction
 * (lower-tile product + element-wise pass), 2 dense SPD inverse (all its launches),
 * 3 per-individual f + gradient, 4 per-individual f only, 5 E-step sums (scatter, or inverse +
 * the two gathers), 6 K_0 build, 7 batched K_i build, 8 batched K_i inverse, 9 gather of the
 * cov blocks (magma_indiv_gather_cov), 10 device-resident per-individual M-step. */
int magma_prof_enable(int on);
int magma_prof_reset(void);
/* Tunables. "resident_max_workgroups": cap on the grid of magma_indiv_mstep_resident (0 = none:
 * one four-wavefront workgroup per compute unit; its wavefronts fill a SIMD's register file, so
 * a cap of G leaves the other compute units to the kernels of other streams -- the theta_0
 * chain). "dag_workgroups": workgroups of the one-launch dense inverse (0 = launch-per-step
 * path). "dedup" (default 1): with identical timestamps and common hyper-parameters the driver
 * inverts K once per evaluation (Computing_functions.R:221-224, 347-350); 0 = batched sum.
 * "resident_help" (default 1): in the device-resident M-step, wavefronts that find no individual
 * left take tile rows of K and column blocks of the product stages of the evaluations still
 * running in their workgroup (identical results; 0 = off; bit 1 = launch every compute unit even
 * for a handful of individuals, for tests). "resident_slow_path" (default 0): see
 * magma_indiv_mstep_resident, status 6.
 * "dag_walker" (default 2): who walks the chain of diagonal tiles of the one-launch inverse -- 2: one
 * workgroup whose wavefronts work a block step ahead of the 64 x 64 factor, 1: one workgroup in
 * lock step, 0: a task per block step. "dag_max_block_rows" (default 256): the one-launch inverse
 * is used up to this many 64-row blocks. Driver switches (default 1): "prefactor" (the M-step's
 * first K_0 inverse started next to the E-step's), "prefetch" (the next E-step's sums started
 * under the monitoring call; inside an EM iteration they start right behind the resident M-step
 * kernel, from the theta_i it leaves in device memory), "speculate" (the next E-step's dense part
 * queued behind an EM iteration's monitoring evaluation, into a second pair of buffers; taken by
 * the next E-step only if it is called with exactly those hyper-parameters and prior mean -- the
 * resident mean / cov of the iteration that queued it are not touched), "resident_gather" (the resident M-step kernel gathers each individual's packed cov block
 * itself instead of a magma_indiv_gather_cov launch between E-step and M-step), "cov_factor" (theta_0 gradient through the E-step's triangular
 * factor of cov; one rank, or every rank with "theta0_shard" 0), "theta0_shard" (several ranks: the
 * two N^3 products of the theta_0 gradient split by block rows and two sums all-reduced per
 * evaluation; 0 = every rank forms the whole gradient, nothing exchanged per evaluation). Results
 * do not depend on any of them except "dag_walker", "cov_factor" and "theta0_shard" (summation
 * order: last bits). */
int magma_set_option(const char* name, long value);
int magma_prof_get(int id, int* h_count, double* h_total_ms);

/* ---- (1) batched SE-kernel covariance ---- */
/* kern_to_cov, tibble branch (CF:85-112): K_i = SE(t_i; theta_i) + sigma_i^2 I for every
 * individual, written column-major at K_out + off2[i] (off2[i] = sum_{j<i} n_j^2). */
int magma_kern_to_cov_batched(int M, const int64_t* off, const int64_t* off2, const double* t,
                              const double* theta, int theta_stride, double* K_out, void* stream);
/* kern_to_cov / kern_to_inv, vector branch (CF:66-81): K_0 on the union grid,
 * diagonal exp(theta1) + sigma^2; A is Np x Np (padding = identity). */
int magma_kern_to_cov_grid(int N, const double* grid, double theta1, double theta2, double sigma,
                           double* A, void* stream);

/* ---- (2) batched small-matrix factorisation / inverse / log-det ---- */
/* kern_to_inv, tibble branch (CF:141-170), plus log|K_i| (dmvnorm's determinant, CF:27):
 * one wavefront per matrix, K_i never written to memory. nmax >= max_i n_i. */
int magma_kern_to_inv_batched(int M, int nmax, const int64_t* off, const int64_t* off2,
                              const double* t, const double* theta, int theta_stride,
                              double* Kinv_out, double* logdet_out, int* info_out, void* stream);

/* ---- (3) E-step ---- */
/* update_inv + update_mean over this rank's individuals (CF:440-478), fused with the
 * construction and inversion of every K_i: P[idx,idx] += K_i^-1 (lower triangle), w[idx] +=
 * K_i^-1 y_i, by FP64 atomics. P (Np x Np) and w (Np) must be zeroed by the caller; partial
 * sums of several ranks are combined by the caller (all-reduce) before magma_estep_finish. */
int magma_estep_scatter(int M, int nmax, const int64_t* off, const double* t, const double* y,
                        const int* idx, const double* theta, int theta_stride, int N, double* P,
                        double* w, int* info_out, void* stream);
/* Rest of e_step (CF:382,386-391) / posterior_mu (MG:118-128): K_0^-1 on the grid with
 * `pen_diag` on the diagonal, P += K_0^-1, w += K_0^-1 m_0, cov = P^-1 (blocked MFMA Cholesky,
 * in place in P, full symmetric), mean = cov w. m0: Np values. h_info: host int, LAPACK-style
 * status of the two factorisations (0 ok; synchronises the stream).
 * reuse_k0inv != 0: the workspace still holds K_0^-1 for exactly (theta1, theta2, pen_diag)
 * from the caller's last f-only magma_theta0_eval on it (what logL_monitoring of the accepted
 * hyper-parameters computed, CF:251-252), so it is not rebuilt.
 * h_f0 (optional, host): logL_GP_mod(theta_0) at the NEW hyper-posterior with this pen_diag,
 * i.e. the theta_0 term of the following logL_monitoring (CF:252), from the same K_0^-1. */
int magma_estep_finish(int N, const double* grid, double theta1, double theta2, double pen_diag,
                       const double* m0, double* P, double* w, double* mean_out, void* work,
                       int reuse_k0inv, double* h_f0, int* h_info, void* stream);
/* The sums of magma_estep_scatter without atomics (update_inv / update_mean, CF:440-478, in a
 * fixed order: an E-step is then repeatable bit for bit). K_i^-1 is materialised in Kinv (sum n_i^2
 * doubles at off2), v = K_i^-1 y_i in v (sum n_i); every element of P / w that receives anything
 * is out[tgt[k]] = sum of val[src[e]], e in [ptr[k], ptr[k+1]) -- lists the host builds once per
 * dataset (any fixed order of sources: it is the order of summation; src indexes Kinv for P and
 * v for w; at most 2^31 sources). P and w must be zero on entry, as for magma_estep_scatter. logdet_scratch: M doubles. */
int magma_estep_gather(int M, int nmax, const int64_t* off, const int64_t* off2, const double* t,
                       const double* y, const double* theta, int theta_stride, double* Kinv,
                       double* v, double* logdet_scratch, int64_t ntgt_P, const int64_t* ptr_P,
                       const int* src_P, const int64_t* tgt_P, int64_t ntgt_w,
                       const int64_t* ptr_w, const int* src_w, const int64_t* tgt_w, double* P,
                       double* w, int* info_out, void* stream);
/* update_inv (CF:440-455) / update_mean (CF:457-478) on EXPLICIT blocks, for callers that hold
 * the K_i^-1 themselves (the fused E-step entry points above never materialise them): individual
 * i's K_i^-1 is n_i x n_i column-major at Kinv + off2[i] (any n_i); P[idx_i, idx_i] += K_i^-1
 * (the whole block, P = prior_inv as a full Np x Np matrix), w[idx_i] += K_i^-1 y_i (w =
 * K_0^-1 m_0 on entry). idx < 0 marks a timestamp that is not on the grid: dropped like
 * `intersect(row.names(...))` drops it (its row/column is not added; K_i^-1 y_i is still formed
 * from the whole block, CF:468). FP64 atomics. */
int magma_update_inv(int M, const int64_t* off, const int64_t* off2, const int* idx,
                     const double* Kinv, int N, double* P, void* stream);
int magma_update_mean(int M, const int64_t* off, const int64_t* off2, const int* idx,
                      const double* Kinv, const double* y, int N, double* w, void* stream);
/* dmvnorm (CF:9-34) with log = TRUE: takes the INVERSE covariance inv_Sigma (Np x Np padded,
 * identity padding; OVERWRITTEN), x and mu (Np values, equal on the padding). h_out =
 * -(n log 2pi - log|inv_Sigma| + z' inv_Sigma z) / 2; h_info != 0 (and NaN) when inv_Sigma is
 * not positive definite. work: magma_workspace_bytes(n). Synchronises. */
int magma_dmvnorm(int n, const double* x, const double* mu, double* inv_Sigma, void* work,
                  double* h_out, int* h_info, void* stream);
/* Dense SPD inverse + log-det on its own (solve(), determinant(): CF:27,133,387). A: in =
 * SPD matrix (lower triangle read), out = full symmetric inverse. d_logdet, d_info: device. */
int magma_spd_inverse(int N, double* A, double* d_logdet, int* d_info, void* work, void* stream);
/* magma_spd_inverse for N <= 1024 WITHOUT a caller workspace and without any allocation. No
 * counterpart in the reference: it exists for callers that cannot allocate device memory -- code
 * under HIP graph capture, tight serving loops (kern_to_inv / solve() of a prediction grid per
 * request), harnesses that forbid hipMalloc. The 24 MiB workspace is a static device array of
 * the library, one copy per GPU, held until the process ends; its device address is looked up
 * once per device on the first call there. Same kernels and results as magma_spd_inverse. Calls
 * that overlap on one GPU (two streams, two threads) share that array and race: serialise them.
 * N > 1024 is rejected (status 2, magma_last_error says why). */
int magma_spd_inverse_static(int N, double* A, double* d_logdet, int* d_info, void* stream);

/* ---- (4) M-step objective and gradient ---- */
/* mode 0: logL_GP_mod + gr_GP_mod per individual (CF:181-201, 304-332) with mean = mean[idx],
 * new_cov = cov[idx, idx]; mode 1: logL_GP + gr_GP (CF:174-179, 279-302: covariance K + C).
 * f_out[M], g_out[3M] (already negated like the reference: these are the gradients of f),
 * info_out[M]. sel (optional): nsel int32 indices of the individuals to evaluate (a lock-step
 * optimiser passes the problems still running); then theta, f_out, g_out and info_out are
 * COMPACT: slot b belongs to individual sel[b]. NULL = all M, slot = individual.
 * cpack: the individuals' blocks C_i = cov[idx_i, idx_i] packed by magma_indiv_gather_cov
 * (the hyper-posterior is fixed during an M-step, so the gather is done once and every
 * evaluation streams contiguous data); NULL = C is zero.
 * The common-HP objective (CF:203-234, 334-368) is the sum of these over individuals with
 * theta_stride = 0.
 * info_out[b] != 0: the Cholesky-type sweep found that slot's covariance not positive definite to
 * rounding. Like the reference (`tryCatch(solve(.), error = ginv(.))`, CF:133, 177, 285) the call
 * does not stop there: f_out / g_out of that slot come from LU with partial pivoting or, when that
 * is singular to machine precision, the pseudo-inverse (MASS::ginv's tolerance), with the
 * log-determinant from the LU of that inverse as in dmvnorm (CF:27). info_out keeps the flag. */
int magma_indiv_eval(int mode, int M, int nmax, const int64_t* off, const double* t,
                     const double* y, const int* idx, const double* theta, int theta_stride,
                     const double* mean, const double* cpack, int want_grad, const int* sel,
                     int nsel, double* f_out, double* g_out, int* info_out, void* stream);
/* new_cov[idx_i, idx_i] of CF:188,312 (`cov[paste0('X', t_i), paste0('X', t_i)]`) for all M
 * individuals at once, cov = the padded N x N matrix left by magma_estep_finish. cpack (device):
 * M * magma_cpack_stride(nmax) doubles; slot i is C_i column-major, zero padded to the tile
 * shape the evaluation kernel for this nmax loads (opaque to the caller). */
int64_t magma_cpack_stride(int nmax);
int magma_indiv_gather_cov(int M, int nmax, const int64_t* off, const int* idx,
                           const double* cov, int N, double* cpack, void* stream);
/* m_step for individual hyper-parameters (CF:419-432: one optim() per individual, mode 0) or
 * train_new_gp's optimisation (MG:93-94, mode 1), the whole loop run natively: M independent
 * L-BFGS-B problems advance in lock step (magma_lbfgsb_*), each round evaluating only the
 * problems still running with one magma_indiv_eval-type launch. h_theta0[3M] (host): starting
 * points. d_theta[3M], d_sel[M], d_f[M], d_g[3M], d_info[M]: device scratch owned by the caller.
 * Host outputs: h_theta_out[3M] (NaN for a problem that could not start), h_f_out[M] (optional),
 * h_f_x0[M] (optional: objective at the starting point, i.e. the individual's term of
 * logL_monitoring at the old hyper-parameters), h_iters, h_nfev, h_status[M] (optional, as
 * magma_lbfgsb_result), h_rounds, h_nevals (optional: launches and evaluations). Synchronises. */
int magma_indiv_mstep(int mode, int M, int nmax, const int64_t* off, const double* t,
                      const double* y, const int* idx, const double* mean, const double* cpack,
                      const double* h_theta0, int m, double factr, double pgtol, int maxit,
                      double* d_theta, int* d_sel, double* d_f, double* d_g, int* d_info,
                      double* h_theta_out, double* h_f_out, double* h_f_x0, int* h_iters,
                      int* h_nfev, int* h_status, int* h_rounds, int64_t* h_nevals, void* stream);
/* The same M problems solved without leaving the GPU: the wavefront that evaluates an
 * individual also steps its L-BFGS-B (the code magma_lbfgsb_* runs on the host, compiled for
 * the device), evaluation after evaluation, until that problem stops; no round trips, no lock
 * step. All pointers are DEVICE memory: theta0[3M] in; theta_out[3M], f_out[M], f_x0[M],
 * iters[M], nfev[M], status[M] out (same meaning as above). order (optional): a permutation of
 * 0..M-1, the order in which individuals are started -- the launch lasts as long as its longest
 * chain of evaluations, so start the expected longest first (e.g. by the previous EM
 * iteration's nfev); results do not depend on it. m <= 5. Asynchronous on stream.
 * status 6 (this entry point only): the covariance at the problem's START point is not positive
 * definite to rounding. The solve() / ginv() evaluation the reference makes there (CF:133) is not
 * inside this kernel: solve such problems with magma_indiv_mstep (magma_m_step does that by
 * itself). A flagged TRIAL point inside a line search gives a non-finite value and the search
 * bisects towards its best point -- a deliberate deviation: the reference's value there is
 * +37 per null direction plus rounding noise of the pseudo-inverse's determinant and only serves
 * to reject the step; option "resident_slow_path" = 1 turns those into status 6 as well. */
int magma_indiv_mstep_resident(int mode, int M, int nmax, const int64_t* off, const double* t,
                               const double* y, const int* idx, const double* mean,
                               const double* cpack, const double* theta0, const int* order,
                               int m, double factr, double pgtol, int maxit, double* theta_out,
                               double* f_out, double* f_x0, int* iters, int* nfev, int* status,
                               void* stream);
/* logL_GP_mod + gr_GP_mod at theta_0 (CF:181-201, 304-332 with length(hp) == 2: sigma :=
 * pen_diag, two gradients): db = hyper-posterior mean on the grid, mean = m0, new_cov = cov.
 * h_out3 = (f, df/dtheta1, df/dtheta2) on the HOST; h_info as above. Synchronises. */
int magma_theta0_eval(int N, const double* grid, double theta1, double theta2, double pen_diag,
                      const double* mean, const double* m0, const double* cov, int want_grad,
                      void* work, double* h_out3, int* h_info, void* stream);
/* m_step's theta_0 problem (CF:412-413: opm(old theta_0, logL_GP_mod, gr_GP_mod, db = mean,
 * mean = m_0, kern_0, new_cov = cov, pen_diag, method "L-BFGS-B")) run natively: the L-BFGS-B of
 * magma_lbfgsb_* (one problem, two parameters) around magma_theta0_eval, without a trip through
 * the host language per evaluation. h_theta0[2]: starting point. Host outputs: h_theta_out[2],
 * and optionally h_f_out (objective at the result), h_f_x0 (objective at the starting point),
 * h_iters, h_nfev, h_status (as magma_lbfgsb_result). A trial point whose K_0 is not positive
 * definite counts as a failed evaluation (f = NaN). Synchronises; the workspace's K_0^-1 is that
 * of the last trial point afterwards. */
int magma_theta0_mstep(int N, const double* grid, const double* h_theta0, double pen_diag,
                       const double* mean, const double* m0, const double* cov, int m,
                       double factr, double pgtol, int maxit, void* work, double* h_theta_out,
                       double* h_f_out, double* h_f_x0, int* h_iters, int* h_nfev, int* h_status,
                       void* stream);
/* Sum of n doubles (deterministic order) -> h_out on the host. Synchronises. */
int magma_sum(int64_t n, const double* x, void* work, double* h_out, void* stream);

/* ---- context / dataset / driver: ONE native call per reference function ----
 * The EM driver of the reference lives in R (`training`, MG:15-82, calling `e_step`, `m_step`,
 * `logL_monitoring`, CF:371-437, 236-264). These entry points run it inside the library, so a
 * host language binds `magma_training` (or one call per reference function) instead of a call per
 * kernel; the device-resident data, streams, the theta_0 side stream and the communicator stay
 * behind two opaque handles.
 *  - magma_ctx: one GPU of one rank: device, a main and a high-priority side stream, and the
 *    communicator of a multi-GPU job (one context per GPU; one host thread uses a context at a
 *    time). world == 1 needs no communicator. For world > 1 either RCCL (magma_comm_unique_id on
 *    one rank, the 128 bytes passed to every rank by the host's own means, then
 *    magma_ctx_create_rccl on each rank with its device current: ncclCommInitRank inside;
 *    librccl is opened with dlopen, not linked) or a caller-supplied all-reduce callback (sum over
 *    ranks, in place, of `count` doubles in DEVICE memory; called with the data complete, may
 *    block; return 0 = ok) -- used by the tests to run several ranks as threads on one GPU.
 *  - magma_dataset: this rank's shard of a db (individuals row0 .. row0 + M - 1 of M_total) on a
 *    union grid of N points, uploaded once, with every buffer the EM needs (precision / cov,
 *    dense workspace, the gather lists of the deterministic E-step, packed cov blocks, optimiser
 *    outputs). Host arrays in: ragged off[M + 1] (off[0] = 0), t, y (timestamps strictly ascending
 *    within an individual, every one of them present in grid), grid ascending. Individuals with
 *    n_i <= 64 run on the one-wavefront-per-individual kernels; larger ones (any n_i <= N) take
 *    the dense route inside the driver: K_i, its inverse and logL_GP_mod / gr_GP_mod through the
 *    N x N kernels on the individual's own timestamps, one individual after the other (meant for
 *    a few long series among many short ones, not for thousands of them).
 * Hyper-parameters are HOST arrays: theta_0 = 2 doubles; theta_i = (theta1, theta2, sigma) rows of
 * ALL M_total individuals (replicated on every rank; pen_diag averages over all of them, CF:381,
 * 409), or one triple with theta_stride 0. Calls are blocking; results in host memory. */
typedef struct magma_ctx magma_ctx;
typedef struct magma_dataset magma_dataset;
typedef int (*magma_allreduce_fn)(void* user, double* dev_buf, int64_t count, void* stream);
typedef struct {
  int64_t theta0_evals; /* dense theta_0 evaluations (M-step + monitoring) */
  int64_t indiv_rounds; /* longest per-individual evaluation chain, summed over M-steps */
  int64_t indiv_evals;  /* per-individual objective + gradient evaluations on this rank */
  int64_t common_evals; /* evaluations of the common-HP objective */
  int64_t flagged;      /* individuals whose K_i was not positive definite in an E-step */
  int64_t slow_resolved; /* individual M-step problems solved again through solve() / ginv() */
} magma_em_stats;

int magma_comm_unique_id(void* h_out128);
int magma_ctx_create(magma_ctx** out, int device);
int magma_ctx_create_rccl(magma_ctx** out, int device, int rank, int world, const void* h_unique_id128);
int magma_ctx_create_callback(magma_ctx** out, int device, int rank, int world,
                              magma_allreduce_fn fn, void* user);
void magma_ctx_destroy(magma_ctx* ctx);
int magma_ctx_info(const magma_ctx* ctx, int* h_rank, int* h_world, int* h_device);
void* magma_ctx_stream(const magma_ctx* ctx); /* the main stream (hipStream_t) */
/* No counterpart in the reference (R is single-process): sends a known vector through the context's
 * communicator on both of the library's streams and checks the sums; the streams are polled against
 * `timeout_ms` rather than synchronised, so that a collective that never completes is status 211
 * (report and exit: the stream stays blocked) and a wrong sum 212. 0 without a communicator. Every
 * rank must call it (it is collective). bench.py runs it before it times anything. */
int magma_ctx_selfcheck(magma_ctx* ctx, int timeout_ms);

int magma_dataset_create(magma_ctx* ctx, magma_dataset** out, int64_t M, int64_t M_total,
                         int64_t row0, const int64_t* h_off, const double* h_t, const double* h_y,
                         int N, const double* h_grid);
void magma_dataset_destroy(magma_dataset* ds);
int magma_dataset_info(const magma_dataset* ds, int64_t* h_M, int* h_nmax, int* h_N, int* h_Np);
int magma_dataset_grid(const magma_dataset* ds, double* h_grid /* N */);
/* device pointers of the resident results, for hosts that chain the kernel-level entry points:
 * "mean" (Np), "cov" (Np x Np), "m0", "grid", "w", "work"; NULL for an unknown name. Valid until
 * the NEXT driver call on this dataset only: an E-step may swap "cov" / "w" with the buffers of
 * its prefetched sums (which the following EM iteration overwrites asynchronously), so ask again
 * after every driver call instead of caching them; writing into "work" discards the K_0^-1 a
 * following E-step would have reused (it is recomputed, results do not change). */
void* magma_dataset_buffer(const magma_dataset* ds, const char* name);
int magma_get_mean(magma_ctx* ctx, magma_dataset* ds, double* h_mean /* N */);
int magma_get_cov(magma_ctx* ctx, magma_dataset* ds, double* h_cov /* N x N, ld */, int64_t ld);
/* load a hyper-posterior computed elsewhere (m_step / logL_* called on their own) */
int magma_set_mean_cov(magma_ctx* ctx, magma_dataset* ds, const double* h_mean, const double* h_cov,
                       int64_t ld);
/* the prior mean m_0 (n_m0 = 1: scalar, or N values) for m_step / logL_monitoring called on a
 * hyper-posterior loaded with magma_set_mean_cov (magma_e_step sets it itself) */
int magma_set_m0(magma_ctx* ctx, magma_dataset* ds, const double* h_m0, int n_m0);
int magma_get_stats(const magma_dataset* ds, magma_em_stats* h_out, int reset);

/* e_step (CF:371-395) on the dataset's grid with pen_diag given (CF:381: mean sigma_i; posterior_mu
 * passes its own, MG:116): mean / cov stay resident. h_m0: n_m0 = 1 (scalar prior mean) or N values.
 * h_info: 0 ok, > 0 K_0 not positive definite, < 0 the hyper-posterior precision; h_f0 (optional):
 * logL_GP_mod(theta_0) at the new hyper-posterior, the theta_0 term of logL_monitoring (CF:252). */
int magma_e_step(magma_ctx* ctx, magma_dataset* ds, const double* h_theta0, const double* h_theta_i,
                 int theta_stride, double pen_diag, const double* h_m0, int n_m0, int* h_info,
                 double* h_f0);
/* m_step (CF:397-437) against the resident hyper-posterior: theta_0 by L-BFGS-B on the side stream
 * (its gradient's N^3 products split over the ranks), the theta_i problems inside one resident
 * kernel (individual HPs) or one 3-parameter problem on the summed objective (common_hp). h_opt4 =
 * (lmm, factr, pgtol, maxit) or NULL for optim's defaults (5, 1e7, 0, 100). Outputs: theta_0 (2),
 * theta_i of ALL M_total individuals (3 M_total, gathered over the ranks). */
int magma_m_step(magma_ctx* ctx, magma_dataset* ds, const double* h_theta0, const double* h_theta_i,
                 int common_hp, const double* h_opt4, double* h_theta0_out, double* h_theta_i_out);
/* logL_monitoring (CF:236-264) at the given hyper-parameters against the resident hyper-posterior */
int magma_logl_monitoring(magma_ctx* ctx, magma_dataset* ds, const double* h_theta0,
                          const double* h_theta_i, int theta_stride, double* h_out);
/* Sum over ALL individuals (every rank's shard, all-reduced) of logL_GP_mod and, with want_grad, of
 * gr_GP_mod against the resident hyper-posterior: logL_GP_mod_common_hp / gr_GP_mod_common_hp
 * (CF:203-234, 334-368) with theta_stride 0 (one triple; identical timestamps then invert once,
 * CF:221-224), the individual part of logL_monitoring (CF:254-261) with theta_stride 3.
 * h_out4 = (sum f, sum g1, sum g2, sum g3). */
int magma_sum_fi(magma_ctx* ctx, magma_dataset* ds, const double* h_theta_i, int theta_stride,
                 int want_grad, double* h_out4);
/* Body of training's loop (MG:32-75): e_step, m_step, logL_monitoring(hp), logL_monitoring(new hp)
 * -> h_logl2 (NaN, NaN when the M-step produced a NaN hyper-parameter, MG:46-51). A failed
 * factorisation in the E-step is an error (status 5). */
int magma_em_iteration(magma_ctx* ctx, magma_dataset* ds, const double* h_theta0,
                       const double* h_theta_i, const double* h_m0, int n_m0, int common_hp,
                       const double* h_opt4, double* h_theta0_out, double* h_theta_i_out,
                       double* h_logl2);
/* training (MG:15-82): at most n_loop_max (reference: 25) EM iterations from ini_hp, stopping rule
 * eps < 1e-3 (MG:64-71). h_trace3 (optional, 3 n_loop_max doubles): per iteration logLL(hp),
 * logLL(new hp), eps -- what the reference prints at MG:56,65. The last E-step's mean / cov stay
 * resident (`param`, MG:78). */
int magma_training(magma_ctx* ctx, magma_dataset* ds, const double* h_m0, int n_m0,
                   const double* h_ini_theta0, const double* h_ini_theta_i3, int common_hp,
                   int n_loop_max, const double* h_opt4, double* h_theta0_out,
                   double* h_theta_i_out, int* h_converged, int* h_iterations, double* h_trace3);
/* posterior_mu (MG:107-133): the dataset on t_pred = sort(unique(db U extra timestamps)) (built
 * here) and the E-step with `pen_diag` on it (MG:116 uses 0.1); *out keeps mean / cov resident
 * for magma_pred_gp_resident / magma_get_*; h_N = size of t_pred (magma_dataset_grid gives it).
 * With world > 1 the grid is the union over the WHOLE db, as in the reference: the ranks exchange
 * their shards' unique timestamps first (two all-reduces), so every rank must make this call (it
 * is collective), with the same h_extra_t or any split of it; a disagreement on the resulting grid
 * is status 2, raised before the precision sums are exchanged. */
int magma_posterior_mu(magma_ctx* ctx, magma_dataset** out, int64_t M, int64_t M_total, int64_t row0,
                       const int64_t* h_off, const double* h_t, const double* h_y, int64_t n_extra,
                       const double* h_extra_t, const double* h_theta0, const double* h_theta_i,
                       int theta_stride, double pen_diag, const double* h_m0_scalar, int* h_N,
                       int* h_info);
/* pred_gp (MG:135-181) for B new individuals against the dataset's resident hyper-posterior: host
 * arrays in and out (ragged like magma_pred_gp_batched; timestamps are matched to the grid here:
 * status 6 = "subscript out of bounds", as MG:161-176 index by name). */
int magma_pred_gp_resident(magma_ctx* ctx, magma_dataset* ds, int B, const int64_t* h_off_n,
                           const double* h_tn, const double* h_yn, const int64_t* h_off_t,
                           const double* h_ts, const double* h_theta, int theta_stride,
                           double* h_mean_out, double* h_var_out, int* h_info);
/* train_new_gp (MG:84-104): 3-parameter L-BFGS-B on logL_GP / gr_GP (covariance K + new_cov) for one
 * new individual: n <= 64 observations (ascending t), the mean process and (optional, NULL = 0)
 * the cov block at its timestamps. h_out3 = (theta1, theta2, sigma); the initial values when the
 * optimiser returns NaN (MG:97-101). */
int magma_train_new_gp(magma_ctx* ctx, int n, const double* h_t, const double* h_y,
                       const double* h_mean, const double* h_cov, int64_t ld_cov,
                       const double* h_ini3, const double* h_opt4, double* h_out3, int* h_status);

/* ---- prediction ---- */
/* pred_gp (MG:172-180) for one new individual with n <= 64 observations and T targets.
 * mean_obs[n], mean_pred[T]: mean process at the observation / target timestamps; idx_n[n],
 * idx_t[T]: their positions in cov's grid (cov NULL = plain GP regression, MG:156-160). */
int magma_pred_gp(int n, const double* tn, const double* yn, const double* mean_obs,
                  const int* idx_n, int T, const double* ts, const double* mean_pred,
                  const int* idx_t, const double* cov, int N, const double* theta3,
                  double* mean_out, double* var_out, int* d_info, void* stream);
/* pred_gp (MG:172-180) for B new individuals against ONE hyper-posterior that stays in HBM
 * (the mean[Np] / cov[Np x Np] that magma_estep_finish left there for posterior_mu, MG:107-133):
 * what `loop_pred` / `full_algo` (simulation_study.R:262-324, MG:203-242) do per new individual,
 * in one launch and without moving cov to the host and back. Ragged inputs: individual b has
 * observations off_n[b]..off_n[b+1] (n_b <= nmax <= 64) and targets off_t[b]..off_t[b+1]
 * (T_b <= tmax); idx_n / idx_t: positions of those timestamps in the grid of mean / cov (every
 * timestamp must be on it, as MG:161-176 index by name). theta: (theta1, theta2, sigma) per
 * individual (theta_stride 3) or shared (0). Outputs ragged like ts; d_info[B] (0 = ok). */
int magma_pred_gp_batched(int B, int nmax, int tmax, const int64_t* off_n, const double* tn,
                          const double* yn, const int* idx_n, const int64_t* off_t,
                          const double* ts, const int* idx_t, const double* mean,
                          const double* cov, int N, const double* theta, int theta_stride,
                          double* mean_out, double* var_out, int* d_info, void* stream);

/* ---- dense helpers exposed for tests and for hosts that compose their own pipelines ---- */
int magma_gemm_nt(int Mr, int Nc, int K, double alpha, const double* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, double* C, int64_t ldc, void* stream);
int magma_copy_pad(int N, const double* src, int64_t ld_src, double* dst, int zero_pad,
                   void* stream);
int magma_copy_unpad(int N, const double* src, double* dst, int64_t ld_dst, void* stream);

/* Test aid, no counterpart in the reference: fills the LDS of every compute unit with NaN bit
 * patterns (workgroups of the maximum LDS size, a few per CU), so that a kernel whose result
 * depends on LDS it never wrote shows up as a NaN in the parity tests instead of once in a
 * while. Synchronises. */
int magma_debug_poison_lds(void* stream);
/* Diagnostic words of the device-resident M-step (24 x uint64; word 0 != 0: a workgroup gave up on
 * a wait between its wavefronts, see small.hip). No counterpart in the reference. */
int magma_debug_resident_diag(unsigned long long* h24);

/* ---- batched L-BFGS-B driver (host side, reverse communication) ----
 * Stands in for `optimr::opm(method = "L-BFGS-B")` -> stats::optim (CF:412-413, 417-418,
 * 426-429; MG:93-94) with its defaults (lmm 5, factr 1e7, pgtol 0, maxit 100, no bounds),
 * run in lock-step over B independent problems of dimension d so that one objective launch
 * serves every problem. ask: fills h_x[B*d] with the points to evaluate and h_active[B];
 * returns the number of problems still running in *h_nactive. tell: takes f[B], g[B*d] at
 * those points (inactive entries ignored; non-finite f counts as +inf). */
typedef struct magma_lbfgsb magma_lbfgsb;
magma_lbfgsb* magma_lbfgsb_create(int B, int d, const double* h_x0, int m, double factr,
                                  double pgtol, int maxit);
int magma_lbfgsb_ask(magma_lbfgsb* s, double* h_x, unsigned char* h_active, int* h_nactive);
int magma_lbfgsb_tell(magma_lbfgsb* s, const double* h_f, const double* h_g);
/* The same with compact I/O, O(running problems) per call: h_sel[k] = index of the k-th running
 * problem (ascending), h_x[k*d..] its trial point; tell takes values in that order. */
int magma_lbfgsb_ask_compact(magma_lbfgsb* s, int* h_sel, double* h_x, int* h_nactive);
int magma_lbfgsb_tell_compact(magma_lbfgsb* s, int n, const int* h_sel, const double* h_f,
                              const double* h_g);
int magma_lbfgsb_result(magma_lbfgsb* s, double* h_x, double* h_f, int* h_iters, int* h_nfev,
                        int* h_status);
void magma_lbfgsb_destroy(magma_lbfgsb* s);

#ifdef __cplusplus
}
#endif
#endif /* MAGMA_HIP_H */
