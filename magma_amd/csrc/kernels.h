// Internal launcher prototypes (device pointers, explicit stream). See small.hip / dense.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magma {

// ---- per-individual (one wavefront per matrix) ----
int launch_se_cov_batched(int M, const int64_t* off, const int64_t* off2, const double* t,
                          const double* theta, int tstride, double* out, hipStream_t st);
int launch_se_inv_batched(int M, int nmax, const int64_t* off, const int64_t* off2,
                          const double* t, const double* theta, int tstride, double* out,
                          double* logdet, int* info, const double* y, double* v, hipStream_t st);
int launch_estep_scatter(int M, int nmax, const int64_t* off, const double* t, const double* y,
                         const int* idx, const double* theta, int tstride, double* P,
                         int64_t ldp, double* w, int* info, hipStream_t st);
int launch_indiv_eval(int nsel, const int* sel, int nmax, const int64_t* off, const double* t,
                      const double* y, const int* idx, const double* theta, int tstride,
                      const double* mean, const double* cpack, int mode, int want_grad,
                      double* f, double* g, int* info, hipStream_t st);
int launch_indiv_mstep(int M, int nmax, const int* order, const int64_t* off, const double* t,
                       const double* y,
                       const int* idx, const double* mean, const double* cpack, int mode,
                       const double* theta0, int lm, double factr, double pgtol, int maxit,
                       double* theta_out, double* f_out, double* fx0_out, int* iters, int* nfev,
                       int* status, hipStream_t st);
void set_resident_max_workgroups(int v);         // process default (option "resident_max_workgroups")
void set_resident_max_workgroups_thread(int v);  // the calling thread's launches; -1 = the default
void set_resident_help(int v);
void set_panel_blocks(int v);
void set_resident_gather_thread(const double* cov, int64_t ld);
bool resident_gather_enabled();
void set_resident_gather(bool on);
void set_resident_slow_path(int v);
int resident_diag_read(unsigned long long* h24);
int launch_indiv_gather_cov(int M, int nmax, const int64_t* off, const int* idx,
                            const double* cov, int64_t ldc, double* cpack, hipStream_t st);
// solve() / ginv() slow path for the individuals whose info != 0 (see small.hip)
int launch_small_slowpath(int M, const int64_t* off, const int64_t* off2, const double* t,
                          const double* y, const int* idx, const double* theta, int tstride,
                          const int* info, double* Kinv, double* v, double* logdet, double* P,
                          int64_t ldp, double* w, int* route, hipStream_t st);
int launch_csr_gather_sum(int64_t ntgt, const int64_t* ptr, const int* src, const int64_t* tgt,
                          const double* val, double* out, int wave_per_target, hipStream_t st);
// update_inv / update_mean on explicit K_i^-1 blocks (any n_i): P[idx_i, idx_i] += K_i^-1 (P != null)
// and / or w[idx_i] += K_i^-1 y_i (w != null), FP64 atomics.
int launch_update_inv(int M, const int64_t* off, const int64_t* off2, const int* idx,
                      const double* Kinv, const double* y, double* P, int64_t ldp, double* w,
                      hipStream_t st);
int launch_pred_gp(int n, const double* tn, const double* yn, const double* mean_obs,
                   const int* idx_n, int T, const double* ts, const double* mean_pred,
                   const int* idx_t, const double* cov, int64_t ldc, const double* theta,
                   double* out_mean, double* out_var, int* info, hipStream_t st);
int launch_pred_gp_batched(int B, int nmax, int tmax, const int64_t* off_n, const double* tn,
                           const double* yn, const int* idx_n, const int64_t* off_t,
                           const double* ts, const int* idx_t, const double* mean_grid,
                           const double* cov, int64_t ldc, const double* theta, int tstride,
                           double* out_mean, double* out_var, int* info, hipStream_t st);

// ---- dense union-grid (N x N, padded to a multiple of 64, column-major, ld = Np) ----
int launch_se_cov_grid(int N, int Np, const double* grid, double th1, double th2, double sigma,
                       double* A, hipStream_t st);
size_t spd_inverse_workspace_doubles(int Np);
// A: in = SPD (lower triangle read), out = full symmetric inverse. logdet/info: device scalars.
int launch_spd_inverse(int Np, double* A, double* logdet, int* info, double* work,
                       hipStream_t st);
// The same as ONE launch of persistent workgroups walking the tile dependency graph
// (dense_dag.hip); used by launch_spd_inverse when enabled ("dag_workgroups" > 0) and Np^2
// doubles fit 32-bit byte offsets.
bool spd_inverse_dag_supported(int Np);
void set_dag_workgroups(int v);
void set_dag_walker(int v);
void set_dag_max_block_rows(int v);  // the one-launch inverse is used up to this many block rows
void set_dag_grid_once(int v);  // this thread's next tile-graph launch only
void set_dag_stamps(void* p);  // diagnostic builds only (-DMAGMA_DAG_STAMPS)
int launch_spd_inverse_dag(int Np, double* A, double* logdet, int* info, double* work,
                           hipStream_t st);
int launch_gemm_nt(int Mr, int Nc, int K, double alpha, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                   hipStream_t st);
int launch_symv(int Np, const double* A, const double* x, double* yout, hipStream_t st);
int launch_axpy(int64_t n, double alpha, const double* x, double* yio, hipStream_t st);
int launch_dot(int64_t n, const double* a, const double* b, double* partial, double* out,
               hipStream_t st);
struct Theta0Shard;
int launch_gemm_nt_rows(int Mr, int Nc, int K, const double* A, int64_t lda, const double* B,
                        int64_t ldb, double* C, int64_t ldc, int rank, int world, hipStream_t st);
int launch_theta0_contract(int N, int Np, const double* T, const double* Kinv,
                           const double* alpha, const double* grid, double th1, double th2,
                           double* Z, double* partials, double* out2, hipStream_t st,
                           const Theta0Shard* sh = nullptr, const double* Bprod = nullptr);
int launch_gemm_nt_tri(int Np, const double* A, const double* B, double* C, hipStream_t st);
int launch_transpose_upper(int Np, const double* U, double* V, hipStream_t st);
int theta0_vec_blocks(int Np);
int launch_theta0_vec(int N, int Np, const double* Kinv, const double* cov, const double* mean,
                      const double* m0, double* alpha, double* vpart, hipStream_t st);
int launch_theta0_finish2(int N, int Np, const double* logdet, const int* info,
                          const double* vpart, const double* pairs, int want_grad, double* out4,
                          hipStream_t st);
int launch_fill(int64_t n, double v, double* x, hipStream_t st);
int launch_poison_lds(double* sink, hipStream_t st);
int launch_set_identity_pad(int N, int Np, double* A, hipStream_t st);
int launch_copy_pad(int N, int Np, const double* src, int64_t lds, double* dst, int zero_pad,
                    hipStream_t st);
int launch_copy_unpad(int N, int Np, const double* src, double* dst, int64_t ldd,
                      hipStream_t st);
int launch_theta0_finish(int N, int Np, const double* scal, double* out3, int want_grad,
                         hipStream_t st);

}  // namespace magma
