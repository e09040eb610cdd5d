// Internal pieces shared by capi.hip and driver.hip (the native EM driver).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magma {

// Split of the theta_0 gradient's two N^3 products over ranks: rank r forms the
// block rows i = r, r + world, ... of T = K_0^-1 C and the lower tiles (i, j <= i) of Z = T K_0^-1
// in those rows (plist: their triangular indices, device memory), contracts them, and the two
// gradient sums are all-reduced. The inverse and the O(N^2) parts are replicated.
struct Theta0Shard {
  int rank, world;
  const int* plist;
  int nplist;
  int (*allreduce)(void* user, double* dev, int64_t n, hipStream_t st);
  void* user;
};

int theta0_eval_impl(int N, const double* grid, double theta1, double theta2, double pen_diag,
                     const double* mean, const double* m0, const double* cov, int want_grad,
                     void* work, double* h_out3, int* h_info, hipStream_t st, const Theta0Shard* sh,
                     const double* cov_fac = nullptr, bool have_inverse = false,
                     double* h_defer4 = nullptr);
// h_defer4 (optional, pinned host memory): (f, g1, g2, info) are copied there asynchronously and
// the call returns without waiting; the caller synchronises st before reading them.
// K_0(theta, pen_diag)^-1, log|K_0| and info into `work`, asynchronously on st, with dag_grid
// workgroups for the tile-graph inverse (0 = default): the first evaluation of a theta_0 M-step
// needs nothing from the E-step for this part, so the driver starts it next to the E-step's own
// inverse (two latency-bound inverses side by side cost little more than one) and passes
// have_inverse / first_have_inverse to the evaluation that follows on the same workspace.
int theta0_prefactor_impl(int N, const double* grid, double theta1, double theta2, double pen_diag,
                          void* work, int dag_grid, hipStream_t st);
int theta0_mstep_impl(int N, const double* grid, const double* h_theta0, double pen_diag,
                      const double* mean, const double* m0, const double* cov, int m, double factr,
                      double pgtol, int maxit, void* work, double* h_theta_out, double* h_f_out,
                      double* h_f_x0, int* h_iters, int* h_nfev, int* h_status, hipStream_t st,
                      const Theta0Shard* sh, const double* cov_fac = nullptr,
                      bool first_have_inverse = false);
// cov_fac (optional, one rank only): V = L^-1 of the precision the E-step just inverted, lower
// triangular, cov = V^T V (estep_cov_factor below). The gradient's K_0^-1 cov K_0^-1 is then
// Y Y^T with Y = K_0^-1 V^T -- a triangular product (N^3) and the lower tiles of a symmetric one
// (N^3) instead of a full product (2 N^3) and the lower tiles (N^3).
// After magma_estep_finish on `work`: V <- U^T, U = L^-T as left in the workspace by the inverse.
int estep_cov_factor(int N, void* work, double* V, hipStream_t st);
// magma_estep_finish without the wait: everything queued on st, (f0, info words) copied to the
// pinned h_pin3 (3 doubles) behind it; estep_finish_read decodes them after st was synchronised.
int estep_finish_launch(int N, const double* grid, double theta1, double theta2, double pen_diag,
                        const double* m0, double* P, double* w, double* mean_out, void* work,
                        int reuse_k0inv, int want_f0, double* h_pin3, hipStream_t st);
void estep_finish_read(const double* h_pin3, double* h_f0, int* h_info);
int dag_diag(int N, void* work, int* h_two);

// logL_GP_mod + gr_GP_mod (CF:181-201, 304-332) of ONE individual with more observations than a
// wavefront handles (n > 64), on the dense kernels: the theta_0 evaluation chain on the
// individual's own timestamps (grid: np values, ascending, zero padded) with three
// hyper-parameters (sigma = th3[2], g3 = -sigma tr(W)). yv = outputs (np, zero padded), mvec =
// mean process at the timestamps, C = cov block (np x np, identity padded; see
// launch_gather_block). h_out4 = (f, g1, g2, g3); work: magma_workspace_bytes(n). Synchronises.
int indiv_dense_eval_impl(int n, const double* grid, const double* th3, const double* yv,
                          const double* mvec, const double* C, int want_grad, void* work,
                          double* h_out4, int* h_info, hipStream_t st);
// The lock-step route of the individual problems (magma_indiv_mstep) over all M individuals
// (subset == nullptr) or over the nsub listed ones: results by position in the list, h_theta0 with
// 3 doubles per listed individual. Every evaluation goes through launch_indiv_eval, i.e. flagged
// covariances are answered by solve() / ginv(). d_* : device staging for max(M, nsub) problems.
int indiv_mstep_lockstep(int mode, int M, int nmax, const int64_t* off, const double* t,
                         const double* y, const int* idx, const double* mean, const double* cpack,
                         const double* h_theta0, const int* subset, int nsub, int m, double factr,
                         double pgtol, int maxit, double* d_theta, int* d_sel, double* d_f,
                         double* d_g, int* d_info, double* h_theta_out, double* h_f_out,
                         double* h_f_x0, int* h_iters, int* h_nfev, int* h_status, int* h_rounds,
                         int64_t* h_nevals, hipStream_t st);
int launch_trace(int N, int64_t ld, const double* A, double* out, hipStream_t st);
int launch_gather_block(int n, int np, const int* idx, const double* cov, int64_t ldc,
                        const double* mean, double* C, double* m, hipStream_t st);

// common timestamps + common hyper-parameters (CF:221-224, 347-350): S = (1/M) sum_i r_i r_i'
// (n x n), the packed C' = C + S of the pseudo-individual that stands for all M, and its outputs
// y' = mean at the timestamps (small.hip, dedup_*). y: M x n outputs, idx: the n grid positions.
int launch_dedup_prepare(int M, int n, int nmax, const double* y, const int* idx, const double* mean,
                         const double* cov, int64_t ldc, double* S, double* cpack, double* y_out,
                         hipStream_t st);
bool dedup_enabled();
void set_dedup(bool on);
bool prefetch_enabled();
void set_prefetch(bool on);
bool covfac_enabled();
// several ranks: split the two N^3 products of the theta_0 gradient by block rows (default), or let
// every rank form the whole gradient (no exchange per evaluation; cov_factor then applies)
bool theta0_shard_enabled();
void set_theta0_shard(bool on);
void set_covfac(bool on);
// the next E-step's dense part queued behind an EM iteration's monitoring evaluation (driver.hip)
void set_speculate(bool on);
bool prefactor_enabled();
void set_prefactor(bool on);

// out4 = (sum f, sum g[:,0], sum g[:,1], sum g[:,2]) over M entries, fixed order (g may be null)
int launch_sum4(int64_t M, const double* f, const double* g, double* out4, hipStream_t st);
// lower 64 x 64 tiles of the Np x Np matrix P <-> contiguous buffer (tile (i, j <= i) at
// (i (i + 1) / 2 + j) * 4096); unpack != 0 copies back into P
int launch_pack_lower(int Np, double* P, double* pack, int unpack, hipStream_t st);

}  // namespace magma
