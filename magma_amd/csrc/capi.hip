// C ABI of libmagma_hip.so (see include/magma_hip.h for the contract and reference citations).
#include "../../include/magma_hip.h"
#include "common.h"
#include "kernels.h"
#include "driver.h"

#include <algorithm>
#include <cstdio>
#include <mutex>
#include <utility>
#include <vector>

namespace magma {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return 100 + (int)e;
}

// Carve-up of the caller-provided dense workspace (all offsets in doubles).
struct Work {
  double *A0, *T, *inv, *Z, *z, *alpha, *tmp, *partials, *vpart, *scal;
  int* iinfo;
  Work(void* base, int Np) {
    const size_t n2 = (size_t)Np * Np;
    const int nt = Np / 64;
    double* p = static_cast<double*>(base);
    A0 = p; p += n2;
    T = p; p += n2;
    inv = p; p += 3 * n2;
    Z = p; p += 2 * n2;  // lower tiles of T K_0^-1, up to two k-slices (theta0_contract_slices)
    z = p; p += Np;
    alpha = p; p += Np;
    tmp = p; p += Np;
    partials = p; p += std::max<size_t>(2 * (size_t)nt * nt, 1024);
    vpart = p; p += 2 * ((size_t)Np / 4 + 1);
    scal = p; p += 16;
    iinfo = reinterpret_cast<int*>(p);
  }
  static size_t bytes(int Np) {
    const size_t n2 = (size_t)Np * Np;
    const int nt = Np / 64;
    return sizeof(double) * (7 * n2 + 3 * (size_t)Np + std::max<size_t>(2 * (size_t)nt * nt, 1024) +
                            2 * ((size_t)Np / 4 + 1) + 16) + 64;
  }
};

static inline hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }

// Optional per-kernel-class timing with HIP events on the launching stream (bench.py's
// roofline figures): off by default, no cost when off.
enum ProfId { PROF_GEMM = 0, PROF_CONTRACT, PROF_SPD_INVERSE, PROF_INDIV_EVAL, PROF_INDIV_F,
              PROF_SCATTER, PROF_COV_GRID, PROF_COV_BATCHED, PROF_INV_BATCHED, PROF_GATHER_COV,
              PROF_INDIV_MSTEP, PROF_N };
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[PROF_N];

struct Prof {
  int id;
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  Prof(int id_, hipStream_t st_) : id(id_), st(st_) {
    if (!g_prof_on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = nullptr; return; }
    (void)hipEventRecord(e0, st);
  }
  ~Prof() {
    if (!e0) return;
    (void)hipEventRecord(e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_ev[id].push_back({e0, e1});
  }
};

static int check_small(int M, int nmax) {
  if (M < 0) return fail(2, "M must be >= 0");
  if (nmax < 1 || nmax > MAGMA_NSMALL_MAX)
    return fail(2, "n_i must be in 1..64 (one wavefront per individual)");
  return 0;
}


// logL_GP_mod + gr_GP_mod at theta_0 (magma_theta0_eval); sh != null: the gradient's two N^3
// products split over ranks by block rows, the two gradient sums all-reduced (driver.h).
int theta0_eval_impl(int N, const double* grid, double theta1, double theta2, double pen_diag,
                     const double* mean, const double* m0, const double* cov, int want_grad,
                     void* work, double* h_out3, int* h_info, hipStream_t st, const Theta0Shard* sh,
                     const double* cov_fac, bool have_inverse, double* h_defer4) {
  if (N < 1) return fail(2, "N must be >= 1");
  const int Np = ((N + 63) / 64) * 64;
  Work wk(work, Np);
  int rc;
  if (sh) cov_fac = nullptr;  // (Y Y^T needs every block row of Y: not with sharded rows)
  if (!have_inverse) {  // (else: theta0_prefactor_impl left K_0^-1, log|K_0| and info in `work`)
    {
      Prof pf(PROF_COV_GRID, st);
      if ((rc = launch_se_cov_grid(N, Np, grid, theta1, theta2, pen_diag, wk.A0, st))) return rc;
    }
    Prof pf(PROF_SPD_INVERSE, st);
    if ((rc = launch_spd_inverse(Np, wk.A0, wk.scal + 0, wk.iinfo + 0, wk.inv, st))) return rc;
  }
  // alpha = K_0^-1 (mean - m0), z'alpha and sum(K_0^-1 o cov) in one pass over K_0^-1
  if ((rc = launch_theta0_vec(N, Np, wk.A0, cov, mean, m0, wk.alpha, wk.vpart, st))) return rc;
  if (want_grad) {
    {
      Prof pf(PROF_GEMM, st);
      if (sh)
        rc = launch_gemm_nt_rows(Np, Np, Np, wk.A0, Np, cov, Np, wk.T, Np, sh->rank, sh->world, st);
      else if (cov_fac)
        rc = launch_gemm_nt_tri(Np, wk.A0, cov_fac, wk.T, st);  // Y = K_0^-1 V^T
      else
        rc = launch_gemm_nt(Np, Np, Np, 1.0, wk.A0, Np, cov, Np, 0.0, wk.T, Np, st);
      if (rc) return rc;
    }
    Prof pf(PROF_CONTRACT, st);
    if ((rc = launch_theta0_contract(N, Np, wk.T, wk.A0, wk.alpha, grid, theta1, theta2, wk.Z,
                                     wk.partials, nullptr, st, sh, cov_fac ? wk.T : nullptr)))
      return rc;
  }
  // (f, g1, g2, info) in one device-to-host copy
  if ((rc = launch_theta0_finish2(N, Np, wk.scal + 0, wk.iinfo + 0, wk.vpart, wk.partials,
                                  want_grad, wk.scal + 8, st)))
    return rc;
  if (sh && want_grad && (rc = sh->allreduce(sh->user, wk.scal + 9, 2, st))) return rc;
  if (h_defer4 != nullptr)
    return check_hip(hipMemcpyAsync(h_defer4, wk.scal + 8, 4 * sizeof(double), hipMemcpyDeviceToHost, st),
                     "theta0_eval copy");
  double h4[4] = {0.0, 0.0, 0.0, 0.0};
  MAGMA_CHECK_HIP(hipMemcpyAsync(h4, wk.scal + 8, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipStreamSynchronize(st));
  h_out3[0] = h4[0];
  h_out3[1] = h4[1];
  h_out3[2] = h4[2];
  if (h_info) *h_info = (int)h4[3];
  return 0;
}

int theta0_prefactor_impl(int N, const double* grid, double theta1, double theta2, double pen_diag,
                          void* work, int dag_grid, hipStream_t st) {
  if (N < 1) return fail(2, "N must be >= 1");
  const int Np = ((N + 63) / 64) * 64;
  Work wk(work, Np);
  if (int rc = launch_se_cov_grid(N, Np, grid, theta1, theta2, pen_diag, wk.A0, st)) return rc;
  Prof pf(PROF_SPD_INVERSE, st);
  set_dag_grid_once(dag_grid);
  const int rc = launch_spd_inverse(Np, wk.A0, wk.scal + 0, wk.iinfo + 0, wk.inv, st);
  set_dag_grid_once(0);
  return rc;
}

// Diagnostic words of the last tile-graph inverse run on `work` (late steps of the walker; the
// flag a workgroup gave up on | its block index << 16), for error messages after info = -999.
int dag_diag(int N, void* work, int* h_two) {
  const int Np = ((N + 63) / 64) * 64;
  Work wk(work, Np);
  const int* ctrl = reinterpret_cast<const int*>(wk.inv + 2 * (size_t)Np * Np);
  return check_hip(hipMemcpy(h_two, ctrl + 2, 2 * sizeof(int), hipMemcpyDeviceToHost), "dag_diag");
}

int estep_cov_factor(int N, void* work, double* V, hipStream_t st) {
  const int Np = ((N + 63) / 64) * 64;
  Work wk(work, Np);
  // both inverse routes keep U = L^-T in the second Np^2 block of their workspace
  return launch_transpose_upper(Np, wk.inv + (size_t)Np * Np, V, st);
}

int indiv_dense_eval_impl(int n, const double* grid, const double* th3, const double* yv,
                          const double* mvec, const double* C, int want_grad, void* work,
                          double* h_out4, int* h_info, hipStream_t st) {
  if (n < 1) return fail(2, "n must be >= 1");
  const int Np = ((n + 63) / 64) * 64;
  const size_t n2 = (size_t)Np * Np;
  Work wk(work, Np);
  int rc;
  if ((rc = launch_se_cov_grid(n, Np, grid, th3[0], th3[1], th3[2], wk.A0, st))) return rc;
  if ((rc = launch_spd_inverse(Np, wk.A0, wk.scal + 0, wk.iinfo + 0, wk.inv, st))) return rc;
  if ((rc = launch_theta0_vec(n, Np, wk.A0, C, yv, mvec, wk.alpha, wk.vpart, st))) return rc;
  if (want_grad) {
    if ((rc = launch_gemm_nt(Np, Np, Np, 1.0, wk.A0, Np, C, Np, 0.0, wk.T, Np, st))) return rc;
    if ((rc = launch_theta0_contract(n, Np, wk.T, wk.A0, wk.alpha, grid, th3[0], th3[1], wk.Z,
                                     wk.partials, nullptr, st, nullptr)))
      return rc;
    // tr(W) = a'a + tr(K^-1 C K^-1) - tr(K^-1), tr(K^-1 C K^-1) = sum(T o K^-1) (both symmetric
    // arguments; the identity padding adds one per padded row, taken off on the host). Z is
    // free again after the contraction: scratch of the two dot products.
    if ((rc = launch_dot(Np, wk.alpha, wk.alpha, wk.Z, wk.scal + 12, st))) return rc;
    if ((rc = launch_dot((int64_t)n2, wk.T, wk.A0, wk.Z + 2048, wk.scal + 13, st))) return rc;
    if ((rc = launch_trace(n, Np, wk.A0, wk.scal + 14, st))) return rc;
  }
  if ((rc = launch_theta0_finish2(n, Np, wk.scal + 0, wk.iinfo + 0, wk.vpart, wk.partials, want_grad,
                                  wk.scal + 8, st)))
    return rc;
  double h8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  MAGMA_CHECK_HIP(hipMemcpyAsync(h8, wk.scal + 8, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipStreamSynchronize(st));
  const int info = (int)h8[3];
  const double nanv = __builtin_nan("");
  h_out4[0] = info ? nanv : h8[0];
  h_out4[1] = info ? nanv : h8[1];
  h_out4[2] = info ? nanv : h8[2];
  h_out4[3] = 0.0;
  if (want_grad) {
    const double trW = h8[4] + (h8[5] - (double)(Np - n)) - h8[6];
    h_out4[3] = info ? nanv : -th3[2] * trW;
  }
  if (h_info) *h_info = info;
  return 0;
}

int theta0_mstep_impl(int N, const double* grid, const double* h_theta0, double pen_diag,
                      const double* mean, const double* m0, const double* cov, int m, double factr,
                      double pgtol, int maxit, void* work, double* h_theta_out, double* h_f_out,
                      double* h_f_x0, int* h_iters, int* h_nfev, int* h_status, hipStream_t st,
                      const Theta0Shard* sh, const double* cov_fac, bool first_have_inverse) {
  if (N < 1) return fail(2, "N must be >= 1");
  if (!h_theta0 || !h_theta_out) return fail(2, "theta0_mstep: NULL argument");
  magma_lbfgsb* opt = magma_lbfgsb_create(1, 2, h_theta0, m, factr, pgtol, maxit);
  if (!opt) return fail(2, "theta0_mstep: could not create the optimiser");
  int rc = 0;
  bool first = true;
  for (;;) {
    double x[2];
    unsigned char active = 0;
    int nact = 0;
    if ((rc = magma_lbfgsb_ask(opt, x, &active, &nact)) || nact == 0) break;
    double out3[3] = {0.0, 0.0, 0.0};
    int info = 0;
    if ((rc = theta0_eval_impl(N, grid, x[0], x[1], pen_diag, mean, m0, cov, 1, work, out3, &info, st,
                               sh, cov_fac, first && first_have_inverse)))
      break;
    const double f = (info == 0) ? out3[0] : __builtin_nan("");  // not SPD: as a failed evaluation
    if (first && h_f_x0) *h_f_x0 = f;
    first = false;
    if ((rc = magma_lbfgsb_tell(opt, &f, out3 + 1))) break;
  }
  if (!rc) rc = magma_lbfgsb_result(opt, h_theta_out, h_f_out, h_iters, h_nfev, h_status);
  magma_lbfgsb_destroy(opt);
  return rc;
}

}  // namespace magma

using namespace magma;

extern "C" {

int magma_hip_version(void) { return 100; }

const char* magma_last_error(void) { return g_err.c_str(); }

int magma_pad(int N) { return ((N + 63) / 64) * 64; }

size_t magma_workspace_bytes(int N) { return Work::bytes(magma_pad(N)); }

int magma_device_count(int* h_count) { return check_hip(hipGetDeviceCount(h_count), "hipGetDeviceCount"); }

int magma_set_device(int device) { return check_hip(hipSetDevice(device), "hipSetDevice"); }

int magma_dev_malloc(void** h_ptr, size_t bytes) { return check_hip(hipMalloc(h_ptr, bytes), "hipMalloc"); }

int magma_dev_free(void* ptr) { return check_hip(hipFree(ptr), "hipFree"); }

int magma_memcpy_h2d(void* dst, const void* h_src, size_t bytes, void* stream) {
  return check_hip(hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, S(stream)), "memcpy h2d");
}

int magma_memcpy_d2h(void* h_dst, const void* src, size_t bytes, void* stream) {
  return check_hip(hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, S(stream)), "memcpy d2h");
}

int magma_memset_zero(void* dst, size_t bytes, void* stream) {
  return check_hip(hipMemsetAsync(dst, 0, bytes, S(stream)), "memset");
}

int magma_stream_sync(void* stream) { return check_hip(hipStreamSynchronize(S(stream)), "stream sync"); }

int magma_kern_to_cov_batched(int M, const int64_t* off, const int64_t* off2, const double* t,
                              const double* theta, int theta_stride, double* K_out, void* stream) {
  if (M < 0) return fail(2, "M must be >= 0");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  Prof pf(PROF_COV_BATCHED, S(stream));
  return launch_se_cov_batched(M, off, off2, t, theta, theta_stride, K_out, S(stream));
}

int magma_kern_to_cov_grid(int N, const double* grid, double theta1, double theta2, double sigma,
                           double* A, void* stream) {
  if (N < 1) return fail(2, "N must be >= 1");
  return launch_se_cov_grid(N, magma_pad(N), grid, theta1, theta2, sigma, A, S(stream));
}

int magma_kern_to_inv_batched(int M, int nmax, const int64_t* off, const int64_t* off2,
                              const double* t, const double* theta, int theta_stride,
                              double* Kinv_out, double* logdet_out, int* info_out, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  Prof pf(PROF_INV_BATCHED, S(stream));
  if (int rc = launch_se_inv_batched(M, nmax, off, off2, t, theta, theta_stride, Kinv_out, logdet_out,
                                     info_out, nullptr, nullptr, S(stream)))
    return rc;
  // info != 0: tryCatch(solve(.), error -> MASS::ginv(.)) (CF:161) on the product side
  return launch_small_slowpath(M, off, off2, t, nullptr, nullptr, theta, theta_stride, info_out,
                               Kinv_out, nullptr, logdet_out, nullptr, 0, nullptr, nullptr, S(stream));
}

int magma_estep_scatter(int M, int nmax, const int64_t* off, const double* t, const double* y,
                        const int* idx, const double* theta, int theta_stride, int N, double* P,
                        double* w, int* info_out, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  Prof pf(PROF_SCATTER, S(stream));
  if (int rc = launch_estep_scatter(M, nmax, off, t, y, idx, theta, theta_stride, P, magma_pad(N), w,
                                    info_out, S(stream)))
    return rc;
  // off2 is not needed when the answer goes straight into P / w
  return launch_small_slowpath(M, off, off, t, y, idx, theta, theta_stride, info_out, nullptr, nullptr,
                               nullptr, P, magma_pad(N), w, nullptr, S(stream));
}

int magma_estep_gather(int M, int nmax, const int64_t* off, const int64_t* off2, const double* t,
                       const double* y, const double* theta, int theta_stride, double* Kinv,
                       double* v, double* logdet_scratch, int64_t ntgt_P, const int64_t* ptr_P,
                       const int* src_P, const int64_t* tgt_P, int64_t ntgt_w,
                       const int64_t* ptr_w, const int* src_w, const int64_t* tgt_w, double* P,
                       double* w, int* info_out, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (ntgt_P < 0 || ntgt_w < 0) return fail(2, "estep_gather: negative list length");
  hipStream_t st = S(stream);
  Prof pf(PROF_SCATTER, st);
  int rc;
  if ((rc = launch_se_inv_batched(M, nmax, off, off2, t, theta, theta_stride, Kinv, logdet_scratch,
                                  info_out, y, v, st)))
    return rc;
  if ((rc = launch_small_slowpath(M, off, off2, t, y, nullptr, theta, theta_stride, info_out, Kinv, v,
                                  logdet_scratch, nullptr, 0, nullptr, nullptr, st)))
    return rc;
  // long lists (many individuals on few grid points, e.g. a common grid: ~M sources per element of
  // P) take one wavefront per target, short ones (C3: ~6) one thread
  const double avg_src = ntgt_P > 0 ? 0.5 * (double)M * (double)nmax * (double)nmax / (double)ntgt_P : 0.0;
  if ((rc = launch_csr_gather_sum(ntgt_P, ptr_P, src_P, tgt_P, Kinv, P, avg_src >= 48.0 ? 1 : 0, st))) return rc;
  return launch_csr_gather_sum(ntgt_w, ptr_w, src_w, tgt_w, v, w, 1, st);
}

int magma_spd_inverse(int N, double* A, double* d_logdet, int* d_info, void* work, void* stream) {
  if (N < 1) return fail(2, "N must be >= 1");
  const int Np = magma_pad(N);
  Work wk(work, Np);
  Prof pf(PROF_SPD_INVERSE, S(stream));
  return launch_spd_inverse(Np, A, d_logdet, d_info, wk.inv, S(stream));
}

// ---- the one entry point that allocates nothing (see the header) ----
// Workspace of the dense inverse for N <= MAGMA_STATIC_NMAX, held by every GPU the library runs on
// until the process ends: 3 Np^2 doubles at Np = 1024 = 24 MiB.
#define MAGMA_STATIC_NMAX 1024
static __device__ __align__(256) double scratch[3 * (size_t)MAGMA_STATIC_NMAX * MAGMA_STATIC_NMAX];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
double* g_scratch_ptr[kMaxDevices] = {nullptr};  // per device: each GPU holds its own copy
}  // namespace

int magma_spd_inverse_static(int N, double* A, double* d_logdet, int* d_info, void* stream) {
  if (N < 1) return fail(2, "N must be >= 1");
  if (N > MAGMA_STATIC_NMAX)
    return fail(2, "magma_spd_inverse_static: N exceeds 1024, the size its built-in workspace is "
                   "made for (use magma_spd_inverse with a caller workspace)");
  int dev = 0;
  MAGMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(2, "magma_spd_inverse_static: device index out of range");
  double* ws = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
      // set-up, once per device, with that device current: the array's name is a host-side
      // stand-in, its device address comes from the runtime
      void* p = nullptr;
      MAGMA_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      if (p == nullptr) return fail(3, "magma_spd_inverse_static: no device address for the workspace");
      g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    ws = g_scratch_ptr[dev];
  }
  // same kernels as magma_spd_inverse; they clear what must start at zero (flags, ticket
  // counter, log-det, info: hipMemsetAsync on the stream) and write L, U before reading them
  return launch_spd_inverse(magma_pad(N), A, d_logdet, d_info, ws, S(stream));
}

}  // extern "C"
// The dense part of the E-step, queued on st without waiting: h_pin3 (pinned host memory, 3
// doubles) receives f0 in [0] and the info words of K_0 and of the precision as two ints in [1]
// once st has passed this point (estep_finish_read decodes them).
namespace magma {
int estep_finish_launch(int N, const double* grid, double theta1, double theta2, double pen_diag,
                        const double* m0, double* P, double* w, double* mean_out, void* work,
                        int reuse_k0inv, int want_f0, double* h_pin3, hipStream_t st) {
  if (N < 1) return fail(2, "N must be >= 1");
  const int Np = magma_pad(N);
  const size_t n2 = (size_t)Np * Np;
  Work wk(work, Np);
  int rc;
  if (!reuse_k0inv) {
    {
      Prof pf(PROF_COV_GRID, st);
      if ((rc = launch_se_cov_grid(N, Np, grid, theta1, theta2, pen_diag, wk.A0, st))) return rc;
    }
    Prof pf(PROF_SPD_INVERSE, st);
    if ((rc = launch_spd_inverse(Np, wk.A0, wk.scal + 0, wk.iinfo + 0, wk.inv, st))) return rc;
  }
  if ((rc = launch_axpy((int64_t)n2, 1.0, wk.A0, P, st))) return rc;
  if ((rc = launch_symv(Np, wk.A0, m0, wk.tmp, st))) return rc;
  if ((rc = launch_axpy(Np, 1.0, wk.tmp, w, st))) return rc;
  {
    Prof pf(PROF_SPD_INVERSE, st);
    if ((rc = launch_spd_inverse(Np, P, wk.scal + 5, wk.iinfo + 1, wk.inv, st))) return rc;
  }
  if ((rc = launch_symv(Np, P, w, mean_out, st))) return rc;
  if (want_f0) {
    // logL_GP_mod(theta_0; db = new mean, new_cov = new cov, pen_diag) from the K_0^-1 that is
    // still in the workspace: z'K_0^-1 z, log|K_0| and sum(K_0^-1 o cov) -- no factorisation.
    if ((rc = launch_theta0_vec(N, Np, wk.A0, P, mean_out, m0, wk.alpha, wk.vpart, st))) return rc;
    if ((rc = launch_theta0_finish2(N, Np, wk.scal + 0, wk.iinfo + 0, wk.vpart, wk.partials, 0,
                                    wk.scal + 8, st)))
      return rc;
  }
  // h_pin3[0] = f0; the two info words as ints in h_pin3[1]
  if (want_f0) MAGMA_CHECK_HIP(hipMemcpyAsync(h_pin3, wk.scal + 8, sizeof(double), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipMemcpyAsync(h_pin3 + 1, wk.iinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  return 0;
}
void estep_finish_read(const double* h_pin3, double* h_f0, int* h_info) {
  const int* ii = reinterpret_cast<const int*>(h_pin3 + 1);
  const int i0 = ii[0], i1 = ii[1];
  if (h_info) *h_info = i0 ? i0 : (i1 ? -i1 : 0);
  // either factorisation failed: mean / cov are garbage, and so is an objective value built on them
  if (h_f0) *h_f0 = (i0 || i1) ? __builtin_nan("") : h_pin3[0];
}
}  // namespace magma
extern "C" {

int magma_estep_finish(int N, const double* grid, double theta1, double theta2, double pen_diag,
                       const double* m0, double* P, double* w, double* mean_out, void* work,
                       int reuse_k0inv, double* h_f0, int* h_info, void* stream) {
  hipStream_t st = S(stream);
  thread_local double* pin = nullptr;
  if (!pin) MAGMA_CHECK_HIP(hipHostMalloc((void**)&pin, 3 * sizeof(double), hipHostMallocDefault));
  if (int rc = estep_finish_launch(N, grid, theta1, theta2, pen_diag, m0, P, w, mean_out, work,
                                   reuse_k0inv, h_f0 != nullptr, pin, st))
    return rc;
  MAGMA_CHECK_HIP(hipStreamSynchronize(st));
  estep_finish_read(pin, h_f0, h_info);
  return 0;
}

int magma_update_inv(int M, const int64_t* off, const int64_t* off2, const int* idx,
                     const double* Kinv, int N, double* P, void* stream) {
  if (M < 0) return fail(2, "M must be >= 0");
  if (N < 1) return fail(2, "N must be >= 1");
  return launch_update_inv(M, off, off2, idx, Kinv, nullptr, P, magma_pad(N), nullptr, S(stream));
}

int magma_update_mean(int M, const int64_t* off, const int64_t* off2, const int* idx,
                      const double* Kinv, const double* y, int N, double* w, void* stream) {
  if (M < 0) return fail(2, "M must be >= 0");
  if (N < 1) return fail(2, "N must be >= 1");
  return launch_update_inv(M, off, off2, idx, Kinv, y, nullptr, magma_pad(N), w, S(stream));
}

int magma_dmvnorm(int n, const double* x, const double* mu, double* inv_Sigma, void* work,
                  double* h_out, int* h_info, void* stream) {
  if (n < 1) return fail(2, "n must be >= 1");
  if (!h_out) return fail(2, "dmvnorm: h_out is NULL");
  const int Np = magma_pad(n);
  hipStream_t st = S(stream);
  Work wk(work, Np);
  int rc;
  // z = x - mu (zero on the padding), ssq = z' Sigma^-1 z
  MAGMA_CHECK_HIP(hipMemcpyAsync(wk.z, x, sizeof(double) * Np, hipMemcpyDeviceToDevice, st));
  if ((rc = launch_axpy(Np, -1.0, mu, wk.z, st))) return rc;
  if ((rc = launch_symv(Np, inv_Sigma, wk.z, wk.alpha, st))) return rc;
  if ((rc = launch_dot(Np, wk.z, wk.alpha, wk.partials, wk.scal + 1, st))) return rc;
  // log|Sigma^-1| (CF:27: determinant() of the INVERSE covariance) from the factorisation
  if ((rc = launch_spd_inverse(Np, inv_Sigma, wk.scal + 0, wk.iinfo + 0, wk.inv, st))) return rc;
  double h2[2] = {0.0, 0.0};
  int hinfo = 0;
  MAGMA_CHECK_HIP(hipMemcpyAsync(h2, wk.scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipMemcpyAsync(&hinfo, wk.iinfo, sizeof(int), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipStreamSynchronize(st));
  if (h_info) *h_info = hinfo;
  const double logdetS = -h2[0];
  *h_out = hinfo ? __builtin_nan("")
                 : -((double)n * 1.8378770664093454835606594728112 + logdetS + h2[1]) / 2.0;
  return 0;
}

int magma_indiv_eval(int mode, int M, int nmax, const int64_t* off, const double* t,
                     const double* y, const int* idx, const double* theta, int theta_stride,
                     const double* mean, const double* cpack, int want_grad, const int* sel,
                     int nsel, double* f_out, double* g_out, int* info_out, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (mode != 0 && mode != 1) return fail(2, "mode must be 0 (logL_GP_mod) or 1 (logL_GP)");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (sel != nullptr && (nsel < 0 || nsel > M)) return fail(2, "nsel must be in 0..M");
  const int count = sel ? nsel : M;
  Prof pf(want_grad ? PROF_INDIV_EVAL : PROF_INDIV_F, S(stream));
  return launch_indiv_eval(count, sel, nmax, off, t, y, idx, theta, theta_stride, mean, cpack,
                           mode, want_grad, f_out, g_out, info_out, S(stream));
}

// Pinned staging of magma_indiv_mstep, per host thread, grown on demand and kept.
namespace {
struct MstepStage {
  size_t cap = 0;
  int *sel = nullptr, *sel2 = nullptr;
  double *theta = nullptr, *f = nullptr, *g = nullptr;
  int ensure(size_t M) {
    if (M <= cap) return 0;
    release();
    const size_t c = M + M / 2 + 64;
    if (hipHostMalloc((void**)&sel, c * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&sel2, c * sizeof(int)) != hipSuccess ||
        hipHostMalloc((void**)&theta, 3 * c * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&f, c * sizeof(double)) != hipSuccess ||
        hipHostMalloc((void**)&g, 3 * c * sizeof(double)) != hipSuccess) {
      release();
      return fail(3, "magma_indiv_mstep: pinned staging allocation failed");
    }
    cap = c;
    return 0;
  }
  void release() {
    if (sel) (void)hipHostFree(sel);
    if (sel2) (void)hipHostFree(sel2);
    if (theta) (void)hipHostFree(theta);
    if (f) (void)hipHostFree(f);
    if (g) (void)hipHostFree(g);
    sel = sel2 = nullptr;
    theta = f = g = nullptr;
    cap = 0;
  }
};
thread_local MstepStage g_stage;
}  // namespace

}  // extern "C"
namespace magma {
// The lock-step route over all M individuals (subset == nullptr) or over the nsub listed ones
// (results then by position in the list; h_theta0 holds 3 doubles per listed individual).
int indiv_mstep_lockstep(int mode, int M, int nmax, const int64_t* off, const double* t,
                         const double* y, const int* idx, const double* mean, const double* cpack,
                         const double* h_theta0, const int* subset, int nsub, int m, double factr,
                         double pgtol, int maxit, double* d_theta, int* d_sel, double* d_f,
                         double* d_g, int* d_info, double* h_theta_out, double* h_f_out,
                         double* h_f_x0, int* h_iters, int* h_nfev, int* h_status, int* h_rounds,
                         int64_t* h_nevals, hipStream_t st) {
  if (h_rounds) *h_rounds = 0;
  if (h_nevals) *h_nevals = 0;
  const int B = subset ? nsub : M;
  if (B <= 0) return 0;
  if (int rc = g_stage.ensure((size_t)B)) return rc;
  magma_lbfgsb* opt = magma_lbfgsb_create(B, 3, h_theta0, m, factr, pgtol, maxit);
  if (!opt) return fail(2, "magma_indiv_mstep: bad optimiser parameters");
  int rc = 0, rounds = 0;
  int64_t nevals = 0;
  for (;;) {
    int na = 0;
    magma_lbfgsb_ask_compact(opt, g_stage.sel, g_stage.theta, &na);
    if (na == 0) break;
    const bool all = (subset == nullptr && na == M);
    const int* hsel = g_stage.sel;
    if (subset) {  // problem k of the optimiser is individual subset[k]
      for (int k = 0; k < na; ++k) g_stage.sel2[k] = subset[g_stage.sel[k]];
      hsel = g_stage.sel2;
    }
    if (!all &&
        (rc = check_hip(hipMemcpyAsync(d_sel, hsel, sizeof(int) * na, hipMemcpyHostToDevice, st),
                        "mstep sel h2d")))
      break;
    if ((rc = check_hip(hipMemcpyAsync(d_theta, g_stage.theta, sizeof(double) * 3 * na,
                                       hipMemcpyHostToDevice, st), "mstep theta h2d")))
      break;
    {
      Prof pf(PROF_INDIV_EVAL, st);
      rc = launch_indiv_eval(na, all ? nullptr : d_sel, nmax, off, t, y, idx, d_theta, 3, mean,
                             cpack, mode, 1, d_f, d_g, d_info, st);
    }
    if (rc) break;
    if ((rc = check_hip(hipMemcpyAsync(g_stage.f, d_f, sizeof(double) * na, hipMemcpyDeviceToHost, st),
                        "mstep f d2h")))
      break;
    if ((rc = check_hip(hipMemcpyAsync(g_stage.g, d_g, sizeof(double) * 3 * na,
                                       hipMemcpyDeviceToHost, st), "mstep g d2h")))
      break;
    if ((rc = check_hip(hipStreamSynchronize(st), "mstep sync"))) break;
    if (rounds == 0 && h_f_x0)
      for (int k = 0; k < na; ++k) h_f_x0[g_stage.sel[k]] = g_stage.f[k];
    magma_lbfgsb_tell_compact(opt, na, g_stage.sel, g_stage.f, g_stage.g);
    ++rounds;
    nevals += na;
  }
  if (!rc) magma_lbfgsb_result(opt, h_theta_out, h_f_out, h_iters, h_nfev, h_status);
  magma_lbfgsb_destroy(opt);
  if (h_rounds) *h_rounds = rounds;
  if (h_nevals) *h_nevals = nevals;
  return rc;
}
}  // namespace magma
extern "C" {

int magma_indiv_mstep(int mode, int M, int nmax, const int64_t* off, const double* t,
                      const double* y, const int* idx, const double* mean, const double* cpack,
                      const double* h_theta0, int m, double factr, double pgtol, int maxit,
                      double* d_theta, int* d_sel, double* d_f, double* d_g, int* d_info,
                      double* h_theta_out, double* h_f_out, double* h_f_x0, int* h_iters,
                      int* h_nfev, int* h_status, int* h_rounds, int64_t* h_nevals, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (mode != 0 && mode != 1) return fail(2, "mode must be 0 (logL_GP_mod) or 1 (logL_GP)");
  return indiv_mstep_lockstep(mode, M, nmax, off, t, y, idx, mean, cpack, h_theta0, nullptr, 0, m, factr,
                              pgtol, maxit, d_theta, d_sel, d_f, d_g, d_info, h_theta_out, h_f_out,
                              h_f_x0, h_iters, h_nfev, h_status, h_rounds, h_nevals, S(stream));
}

int magma_indiv_mstep_resident(int mode, int M, int nmax, const int64_t* off, const double* t,
                               const double* y, const int* idx, const double* mean,
                               const double* cpack, const double* theta0, const int* order,
                               int m, double factr, double pgtol, int maxit, double* theta_out,
                               double* f_out, double* f_x0, int* iters, int* nfev, int* status,
                               void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (mode != 0 && mode != 1) return fail(2, "mode must be 0 (logL_GP_mod) or 1 (logL_GP)");
  if (m < 1 || m > 5) return fail(2, "the device-resident optimiser keeps at most 5 correction pairs");
  Prof pf(PROF_INDIV_MSTEP, S(stream));
  return launch_indiv_mstep(M, nmax, order, off, t, y, idx, mean, cpack, mode, theta0, m, factr, pgtol,
                            maxit, theta_out, f_out, f_x0, iters, nfev, status, S(stream));
}

int magma_set_option(const char* name, long value) {
  if (!name) return fail(2, "option name is NULL");
  if (std::string(name) == "prefactor") {  // first theta_0 inverse of an M-step started under the E-step
    set_prefactor(value != 0);
    return 0;
  }
  if (std::string(name) == "theta0_shard") {  // several ranks: theta_0 gradient products split by block rows
    set_theta0_shard(value != 0);
    return 0;
  }
  if (std::string(name) == "cov_factor") {  // theta_0 gradient through the E-step's factor of cov
    set_covfac(value != 0);
    return 0;
  }
  if (std::string(name) == "speculate") {  // next E-step's dense part queued behind the monitoring call
    set_speculate(value != 0);
    return 0;
  }
  if (std::string(name) == "prefetch") {  // next E-step's sums started under the monitoring call
    set_prefetch(value != 0);
    return 0;
  }
  if (std::string(name) == "dedup") {  // CF:221-224: invert once for identical timestamps
    set_dedup(value != 0);
    return 0;
  }
  if (std::string(name) == "resident_max_workgroups") {
    set_resident_max_workgroups((int)value);
    return 0;
  }
  if (std::string(name) == "resident_help") {  // idle wavefronts take column blocks of running ones
    set_resident_help((int)value);
    return 0;
  }
  if (std::string(name) == "resident_gather") {  // driver: packed cov blocks gathered inside the resident kernel
    set_resident_gather(value != 0);
    return 0;
  }
  if (std::string(name) == "resident_slow_path") {  // status 6 on every flagged point, not only the start
    set_resident_slow_path((int)value);
    return 0;
  }
  if (std::string(name) == "panel_blocks") {  // launch-per-step inverse: panels per super-step
    set_panel_blocks((int)value);
    return 0;
  }
  if (std::string(name) == "dag_workgroups") {
    if (value < 0 || value > 4096) return fail(2, "dag_workgroups must be in 0..4096");
    set_dag_workgroups((int)value);
    return 0;
  }
  if (std::string(name) == "dag_max_block_rows") {  // (default 256: the tile graph wins at every size measured, N <= 16,000)
    if (value < 0 || value > 256) return fail(2, "dag_max_block_rows must be in 0..256");
    set_dag_max_block_rows((int)value);
    return 0;
  }
  if (std::string(name) == "dag_walker") {  // one workgroup walks the chain of diagonal tiles
    set_dag_walker((int)value);
    return 0;
  }
  if (std::string(name) == "dag_stamps") {  // device buffer of 8 int64 per ticket; diagnostic builds
    set_dag_stamps(reinterpret_cast<void*>(value));
    return 0;
  }
  return fail(2, std::string("unknown option: ") + name);
}

int64_t magma_cpack_stride(int nmax) {
  if (nmax < 1 || nmax > MAGMA_NSMALL_MAX) return 0;
  return (int64_t)cpack_ldp(nmax) * small_template_nmax(nmax);
}

int magma_indiv_gather_cov(int M, int nmax, const int64_t* off, const int* idx,
                           const double* cov, int N, double* cpack, void* stream) {
  if (int rc = check_small(M, nmax)) return rc;
  if (N < 1) return fail(2, "N must be >= 1");
  Prof pf(PROF_GATHER_COV, S(stream));
  return launch_indiv_gather_cov(M, nmax, off, idx, cov, magma_pad(N), cpack, S(stream));
}

int magma_theta0_eval(int N, const double* grid, double theta1, double theta2, double pen_diag,
                      const double* mean, const double* m0, const double* cov, int want_grad,
                      void* work, double* h_out3, int* h_info, void* stream) {
  return theta0_eval_impl(N, grid, theta1, theta2, pen_diag, mean, m0, cov, want_grad, work, h_out3,
                          h_info, S(stream), nullptr);
}

int magma_theta0_mstep(int N, const double* grid, const double* h_theta0, double pen_diag,
                       const double* mean, const double* m0, const double* cov, int m,
                       double factr, double pgtol, int maxit, void* work, double* h_theta_out,
                       double* h_f_out, double* h_f_x0, int* h_iters, int* h_nfev, int* h_status,
                       void* stream) {
  return theta0_mstep_impl(N, grid, h_theta0, pen_diag, mean, m0, cov, m, factr, pgtol, maxit, work,
                           h_theta_out, h_f_out, h_f_x0, h_iters, h_nfev, h_status, S(stream), nullptr);
}

int magma_debug_resident_diag(unsigned long long* h24) { return magma::resident_diag_read(h24); }

int magma_debug_poison_lds(void* stream) {
  double* sink = nullptr;
  MAGMA_CHECK_HIP(hipMalloc((void**)&sink, sizeof(double)));
  int rc = launch_poison_lds(sink, S(stream));
  if (!rc) rc = check_hip(hipStreamSynchronize(S(stream)), "poison_lds sync");
  (void)hipFree(sink);
  return rc;
}

int magma_sum(int64_t n, const double* x, void* work, double* h_out, void* stream) {
  if (n < 0) return fail(2, "n must be >= 0");
  double* w = static_cast<double*>(work);  // needs >= 1025 doubles
  hipStream_t st = S(stream);
  if (int rc = launch_dot(n, x, nullptr, w, w + 1024, st)) return rc;
  MAGMA_CHECK_HIP(hipMemcpyAsync(h_out, w + 1024, sizeof(double), hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipStreamSynchronize(st));
  return 0;
}

int magma_pred_gp(int n, const double* tn, const double* yn, const double* mean_obs,
                  const int* idx_n, int T, const double* ts, const double* mean_pred,
                  const int* idx_t, const double* cov, int N, const double* theta3,
                  double* mean_out, double* var_out, int* d_info, void* stream) {
  if (n < 1 || n > MAGMA_NSMALL_MAX) return fail(2, "pred_gp: n must be in 1..64");
  if (T < 0) return fail(2, "pred_gp: T must be >= 0");
  if (cov != nullptr && (idx_n == nullptr || idx_t == nullptr))
    return fail(2, "pred_gp: idx_n / idx_t required with cov");
  return launch_pred_gp(n, tn, yn, mean_obs, idx_n, T, ts, mean_pred, idx_t, cov,
                        cov ? magma_pad(N) : 0, theta3, mean_out, var_out, d_info, S(stream));
}

int magma_pred_gp_batched(int B, int nmax, int tmax, const int64_t* off_n, const double* tn,
                          const double* yn, const int* idx_n, const int64_t* off_t,
                          const double* ts, const int* idx_t, const double* mean,
                          const double* cov, int N, const double* theta, int theta_stride,
                          double* mean_out, double* var_out, int* d_info, void* stream) {
  if (B < 0) return fail(2, "pred_gp_batched: B must be >= 0");
  if (nmax < 1 || nmax > MAGMA_NSMALL_MAX) return fail(2, "pred_gp_batched: nmax must be in 1..64");
  if (tmax < 0 || tmax > 8 * 65535) return fail(2, "pred_gp_batched: tmax must be in 0..524280");
  if (B > 65535) return fail(2, "pred_gp_batched: at most 65535 individuals per call");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (N < 1 || !mean || !cov || !idx_n || !idx_t)
    return fail(2, "pred_gp_batched: mean, cov (on one grid of N points) and idx_n / idx_t required");
  return launch_pred_gp_batched(B, nmax, tmax, off_n, tn, yn, idx_n, off_t, ts, idx_t, mean, cov,
                                magma_pad(N), theta, theta_stride, mean_out, var_out, d_info,
                                S(stream));
}

int magma_prof_enable(int on) {
  g_prof_on = on != 0;
  return 0;
}

int magma_prof_reset(void) {
  for (int i = 0; i < PROF_N; ++i) {
    for (auto& pr : g_prof_ev[i]) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    g_prof_ev[i].clear();
  }
  return 0;
}

int magma_prof_get(int id, int* h_count, double* h_total_ms) {
  if (id < 0 || id >= PROF_N) return fail(2, "prof id out of range");
  double tot = 0.0;
  for (auto& pr : g_prof_ev[id]) {
    MAGMA_CHECK_HIP(hipEventSynchronize(pr.second));
    float ms = 0.f;
    MAGMA_CHECK_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
  }
  *h_count = (int)g_prof_ev[id].size();
  *h_total_ms = tot;
  return 0;
}

int magma_gemm_nt(int Mr, int Nc, int K, double alpha, const double* A, int64_t lda,
                  const double* B, int64_t ldb, double beta, double* C, int64_t ldc, void* stream) {
  return launch_gemm_nt(Mr, Nc, K, alpha, A, lda, B, ldb, beta, C, ldc, S(stream));
}

int magma_copy_pad(int N, const double* src, int64_t ld_src, double* dst, int zero_pad,
                   void* stream) {
  return launch_copy_pad(N, magma_pad(N), src, ld_src, dst, zero_pad, S(stream));
}

int magma_copy_unpad(int N, const double* src, double* dst, int64_t ld_dst, void* stream) {
  return launch_copy_unpad(N, magma_pad(N), src, dst, ld_dst, S(stream));
}

}  // extern "C"
