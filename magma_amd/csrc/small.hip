// Per-individual kernels: one wavefront (64 lanes) per individual covariance matrix.
//
// Layout: lane r owns row r of the n_i x n_i matrix in registers a[0..NMAX) (compile-time
// indexed, loops fully unrolled); values that every lane needs (a pivot column, a row of
// K^-1, timestamps) are broadcast through LDS. Matrices smaller than NMAX are padded with the
// identity, which leaves the inverse, the log-determinant and every trace unchanged.
//
// Factorisation: symmetric sweep (Gauss-Jordan on an SPD matrix without pivoting). Its pivots
// are the squares of the Cholesky diagonal (d_k = L_kk^2), so the log-det is sum(log d_k) and
// "not positive definite" is detected exactly where a Cholesky would fail; it produces the
// full inverse in place in n^3 multiply-adds without a separate triangular-inverse pass.
//
// Reference semantics replaced (Computing_functions.R = CF, MAGMA.R = MG):
//   se_cov_batched      : kern_to_cov tibble branch, CF:85-112
//   se_inv_batched      : kern_to_inv tibble branch, CF:141-170 (+ log|K| of dmvnorm, CF:27)
//   estep_scatter       : kern_to_inv + update_inv + update_mean, CF:383,440-478
//   indiv_eval          : logL_GP_mod / gr_GP_mod (CF:181-201, 304-332), logL_GP / gr_GP
//                         (CF:174-179, 279-302)
//   pred_gp             : MG:172-180
#include "common.h"
#include "kernels.h"

namespace magma {

#ifndef MAGMA_INDIV_WAVES
#define MAGMA_INDIV_WAVES 1
#endif

constexpr double LOG_2PI = 1.8378770664093454835606594728112;

// ------------------------------------------------------------------ row build
// a[c] = K[r][c] for the SE kernel: off-diagonal exp(th1 - h (t_r - t_c)^2), diagonal
// exp(th1) + sigma^2 (CF:76,103: as.matrix(dist) has a zero diagonal). Identity padding.
template <int NMAX>
__device__ __forceinline__ void build_row(double (&a)[NMAX], int lane, int n, double tr,
                                          const double* tbuf, const SEParams& p) {
  const bool rv = lane < n;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    double d = tr - tbuf[c];
    double v = exp(p.th1 - p.h * d * d);
    v = (c == lane) ? (p.eth1 + p.sig2) : v;
    a[c] = (rv && c < n) ? v : ((c == lane) ? 1.0 : 0.0);
    // keep the unrolled exp() bodies sequential: interleaving all of them spills registers
    __builtin_amdgcn_sched_barrier(0);
  }
}

// 1/d by v_rcp_f64 + two Newton steps (<= 2 ulp): keeps a full IEEE division off the
// serial pivot chain.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}

// ------------------------------------------------------------------ symmetric sweep
// In: a = row `lane` of SPD matrix A (identity padded). Out: a = row of A^-1.
// Stored row r is kept as true_row / s_r so that the pivot row needs no per-element scaling.
// colbuf: 64 doubles of LDS owned by this wavefront. Returns log|A| and info (0 = SPD,
// k+1 = first non-positive pivot, like LAPACK dpotrf's info).
template <int NMAX>
__device__ __forceinline__ void sweep_inverse(double (&a)[NMAX], int lane, double* colbuf,
                                              double& logdet, int& info) {
  double s = 1.0;
  double pm = 1.0;
  int pe = 0;
  info = 0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    colbuf[lane] = a[k] * s;  // true column k == true row k (symmetry)
    __syncthreads();
    double d = colbuf[k];
    const bool ok = (d > 0.0) && (d < 1.7e308);  // uniform across the wavefront
    info = (!ok && info == 0) ? k + 1 : info;
    d = ok ? d : 1.0;
    int e;
    pm *= frexp(d, &e);
    pe += e;
    const double pinv = fast_rcp(d);
    const bool isk = (lane == k);
    const double f = isk ? 0.0 : a[k] * pinv;
#pragma unroll
    for (int c = 0; c < NMAX; c += 2) {
      double2 b = *reinterpret_cast<const double2*>(&colbuf[c]);
      a[c] = fma(-f, b.x, a[c]);
      a[c + 1] = fma(-f, b.y, a[c + 1]);
    }
    a[k] = isk ? -1.0 : f;
    s = isk ? pinv : s;
    __syncthreads();
  }
  const double ms = -s;  // all pivots swept: stored*s = -A^-1
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    a[c] *= ms;
    // pin the finished row: otherwise hipcc sinks each element's whole update chain into the
    // callers' conditional stores and spills every broadcast value of every step
    asm volatile("" : "+v"(a[c]));
  }
  logdet = log(pm) + (double)pe * 0.69314718055994530941723212145818;
}

// ------------------------------------------------------------------ K1: covariance only
// One workgroup per individual; timestamps staged in LDS, coalesced column-major writes.
__global__ __launch_bounds__(256) void se_cov_batched_kernel(
    int M, const int64_t* __restrict__ off, const int64_t* __restrict__ off2,
    const double* __restrict__ t, const double* __restrict__ theta, int tstride,
    double* __restrict__ out) {
  __shared__ double tbuf[MAGMA_NSMALL_MAX];
  for (int i = blockIdx.x; i < M; i += gridDim.x) {
    const int64_t o = off[i];
    const int n = (int)(off[i + 1] - o);
    const double* th = theta + (int64_t)i * tstride;
    const SEParams p = se_params(th[0], th[1], th[2]);
    __syncthreads();
    if (threadIdx.x < n) tbuf[threadIdx.x] = t[o + threadIdx.x];
    __syncthreads();
    double* K = out + off2[i];
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int r = e % n, c = e / n;
      const double d = tbuf[r] - tbuf[c];
      K[e] = (r == c) ? (p.eth1 + p.sig2) : exp(p.th1 - p.h * d * d);
    }
  }
}

// ------------------------------------------------------------------ K1+K3: inverse to HBM
template <int NMAX>
__global__ __launch_bounds__(64) void se_inv_batched_kernel(
    int M, const int64_t* __restrict__ off, const int64_t* __restrict__ off2,
    const double* __restrict__ t, const double* __restrict__ theta, int tstride,
    double* __restrict__ out, double* __restrict__ logdet_out, int* __restrict__ info_out) {
  __shared__ double tbuf[64];
  __shared__ __attribute__((aligned(16))) double colbuf[64];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)i * tstride;
  const SEParams p = se_params(th[0], th[1], th[2]);
  const double tr = (lane < n) ? t[o + lane] : 0.0;
  tbuf[lane] = tr;
  __syncthreads();
  double a[NMAX];
  build_row<NMAX>(a, lane, n, tr, tbuf, p);
  double logdet;
  int info;
  sweep_inverse<NMAX>(a, lane, colbuf, logdet, info);
  double* Kinv = out + off2[i];
  if (lane < n) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c < n) Kinv[(int64_t)c * n + lane] = a[c];
  }
  if (lane == 0) {
    logdet_out[i] = logdet;
    info_out[i] = info;
  }
}

// ------------------------------------------------------------------ E-step scatter
// P[idx_a, idx_b] += K_i^-1[a, b] (lower triangle only: idx ascending within an individual),
// w[idx_a] += (K_i^-1 y_i)[a]. FP64 hardware atomics into the L2-resident N x N partial
// precision; K_i and K_i^-1 never touch HBM.
template <int NMAX>
__global__ __launch_bounds__(64) void estep_scatter_kernel(
    int M, const int64_t* __restrict__ off, const double* __restrict__ t,
    const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, double* __restrict__ P, int64_t ldp,
    double* __restrict__ w, int* __restrict__ info_out) {
  __shared__ double tbuf[64];
  __shared__ double ybuf[64];
  __shared__ int ibuf[64];
  __shared__ __attribute__((aligned(16))) double colbuf[64];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)i * tstride;
  const SEParams p = se_params(th[0], th[1], th[2]);
  const bool rv = lane < n;
  const double tr = rv ? t[o + lane] : 0.0;
  const int ir = rv ? idx[o + lane] : 0;
  tbuf[lane] = tr;
  ybuf[lane] = rv ? y[o + lane] : 0.0;
  ibuf[lane] = ir;
  __syncthreads();
  double a[NMAX];
  build_row<NMAX>(a, lane, n, tr, tbuf, p);
  double logdet;
  int info;
  sweep_inverse<NMAX>(a, lane, colbuf, logdet, info);
  if (lane == 0) info_out[i] = info;
  if (info != 0) return;  // flagged: contributes nothing; the host reports it
  double wi = 0.0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) wi = fma(a[c], ybuf[c], wi);
  if (rv) {
    unsafeAtomicAdd(&w[ir], wi);
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      if (c <= lane) unsafeAtomicAdd(&P[(int64_t)ibuf[c] * ldp + ir], a[c]);
    }
  }
}

// ------------------------------------------------------------------ M-step evaluation
// mode 0 (logL_GP_mod / gr_GP_mod): K = K(theta, sigma);
//        f = 1/2 (n log 2pi + log|K| + z'K^-1 z) + 1/2 sum(K^-1 o C),
//        W = aa' + K^-1 C K^-1 - K^-1, g = -(1/2 sum W o K_nl, 1/2 sum W o dK2, sigma tr W).
// mode 1 (logL_GP / gr_GP): K = K(theta, sigma) + C; f without the trace term; W = aa' - K^-1.
// C = cov[idx, idx] gathered from the N x N hyper-posterior covariance (cov == nullptr: zero).
// Workgroup b handles individual sel[b] (sel == nullptr: b), so a lock-step optimiser only
// launches the problems that are still running.
// Gradient data flow (one LDS matrix, dynamic): Ks = K^-1 (symmetric: "own element" and
// "broadcast row" layouts coincide). T' = C K^-1 is accumulated in registers with the own
// operand C[r][c] gathered straight from the L2-resident cov, then written over Ks as rows;
// Z = K^-1 T' uses the register-resident row of K^-1 as own operand and broadcast rows of T'.
template <int NMAX>
__global__ __launch_bounds__(64, MAGMA_INDIV_WAVES) void indiv_eval_kernel(
    int nsel, const int* __restrict__ sel, const int64_t* __restrict__ off,
    const double* __restrict__ t, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, const double* __restrict__ mean,
    const double* __restrict__ cov, int64_t ldc, int mode, int want_grad,
    double* __restrict__ f_out, double* __restrict__ g_out, int* __restrict__ info_out) {
  constexpr int LD = NMAX + 2;
  __shared__ double tbuf[64];
  __shared__ double zbuf[64];
  __shared__ double abuf[64];
  __shared__ int ibuf[64];
  __shared__ __attribute__((aligned(16))) double colbuf[64];
  extern __shared__ __attribute__((aligned(16))) double Ks[];

  const int lane = threadIdx.x;
  const int io = blockIdx.x;  // slot of theta / f / g / info (compact when sel is given)
  const int i = sel ? sel[io] : io;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)io * tstride;
  const double sigma = th[2];
  const SEParams p = se_params(th[0], th[1], sigma);
  const bool rv = lane < n;
  const bool lv = lane < NMAX;  // lanes beyond the padded matrix own no row
  const double tr = rv ? t[o + lane] : 0.0;
  const int ir = rv ? idx[o + lane] : 0;
  const double zr = rv ? (y[o + lane] - mean[ir]) : 0.0;
  tbuf[lane] = tr;
  zbuf[lane] = zr;
  ibuf[lane] = ir;
  __syncthreads();

  double a[NMAX];
  build_row<NMAX>(a, lane, n, tr, tbuf, p);
  const bool have_c = (cov != nullptr);
  const double* crow = cov + ir;  // column-major symmetric: C[r][c] = cov[idx_c * ldc + idx_r]
  if (mode == 1 && have_c) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (rv && c < n) a[c] += crow[(int64_t)ibuf[c] * ldc];
  }
  double logdet;
  int info;
  sweep_inverse<NMAX>(a, lane, colbuf, logdet, info);

  // alpha = K^-1 z ; ssq = z' alpha
  double alpha = 0.0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) alpha = fma(a[c], zbuf[c], alpha);
  abuf[lane] = alpha;
  const double ssq = wave_sum(zr * alpha);

  const bool grad_c = want_grad && mode == 0 && have_c;
  double cor = 0.0;
  double z[NMAX];  // row r of K^-1 C K^-1 (zero when there is no C term)
#pragma unroll
  for (int c = 0; c < NMAX; ++c) z[c] = 0.0;
  if (mode == 0 && have_c && !grad_c) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      const double cv = (rv && c < n) ? crow[(int64_t)ibuf[c] * ldc] : 0.0;
      cor = fma(a[c], cv, cor);
    }
  }
  if (grad_c) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (lv) Ks[c * LD + lane] = a[c];
    __syncthreads();
    // T'[r][d] = sum_c C[r][c] K^-1[c][d], trace term on the way; own C prefetched 2 ahead
    double cv0 = (rv && 0 < n) ? crow[(int64_t)ibuf[0] * ldc] : 0.0;
    double cv1 = (rv && 1 < n) ? crow[(int64_t)ibuf[1] * ldc] : 0.0;
    for (int c = 0; c < n; ++c) {
      const double cv = cv0;
      cv0 = cv1;
      cv1 = (rv && c + 2 < n) ? crow[(int64_t)ibuf[c + 2] * ldc] : 0.0;
      const double kown = lv ? Ks[c * LD + lane] : 0.0;
      cor = fma(kown, cv, cor);
      const double* krow = &Ks[c * LD];
#pragma unroll
      for (int d = 0; d < NMAX; d += 2) {
        const double2 kk = *reinterpret_cast<const double2*>(&krow[d]);
        z[d] = fma(cv, kk.x, z[d]);
        z[d + 1] = fma(cv, kk.y, z[d + 1]);
      }
    }
    __syncthreads();
    // rows of T' over Ks (lane d writes row d), then Z[r][b] = sum_d K^-1[r][d] T'[d][b]
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
      if (lv) Ks[lane * LD + b] = z[b];
      z[b] = 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < NMAX; ++d) {
      const double kown = a[d];
      const double* trow = &Ks[d * LD];
#pragma unroll
      for (int b = 0; b < NMAX; b += 2) {
        const double2 tt = *reinterpret_cast<const double2*>(&trow[b]);
        z[b] = fma(kown, tt.x, z[b]);
        z[b + 1] = fma(kown, tt.y, z[b + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);  // no load hoisting across rows: register pressure
    }
  }
  if (mode == 0 && have_c) cor = wave_sum(cor);
  const double f = 0.5 * ((double)n * LOG_2PI + logdet + ssq) + 0.5 * cor;
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  if (lane == 0) {
    f_out[io] = (info == 0) ? f : nanv;
    info_out[io] = info;
  }
  if (!want_grad) return;
  __syncthreads();  // abuf visible
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
  for (int b = 0; b < NMAX; ++b) {
    const double d = tr - tbuf[b];
    const double gq = p.h * d * d;
    double knl = exp(p.th1 - gq);
    knl = (b == lane) ? p.eth1 : knl;
    const double W = fma(alpha, abuf[b], z[b] - a[b]);
    const double wk = (rv && b < n) ? W * knl : 0.0;
    s1 += wk;
    s2 = fma(wk, gq, s2);  // dK/dtheta2 = gq * K_nl, zero on the diagonal (gq = 0 there)
    s3 += (b == lane && rv) ? W : 0.0;
    __builtin_amdgcn_sched_barrier(0);
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  s3 = wave_sum(s3);
  if (lane == 0) {
    g_out[3 * (int64_t)io + 0] = (info == 0) ? -0.5 * s1 : nanv;
    g_out[3 * (int64_t)io + 1] = (info == 0) ? -0.5 * s2 : nanv;
    g_out[3 * (int64_t)io + 2] = (info == 0) ? -sigma * s3 : nanv;
  }
}

// ------------------------------------------------------------------ prediction (MG:172-180)
// One wavefront inverts cov_tn_tn = K(tn) + C[in,in]; each workgroup then handles a chunk
// of target timestamps: Mean = m_t + k' inv (y - m_obs), Var = (e^th1 + sigma^2 + C[t,t])
// - k' inv k, with k = kern(tn - t) + C[in, t] (plain kernel, no noise: MG:174).
template <int NMAX>
__global__ __launch_bounds__(64) void pred_gp_kernel(
    int n, const double* __restrict__ tn, const double* __restrict__ yn,
    const double* __restrict__ mean_obs, const int* __restrict__ idx_n, int T,
    const double* __restrict__ ts, const double* __restrict__ mean_pred,
    const int* __restrict__ idx_t, const double* __restrict__ cov, int64_t ldc,
    const double* __restrict__ theta, double* __restrict__ out_mean,
    double* __restrict__ out_var, int* __restrict__ info_out, int chunk) {
  __shared__ double tbuf[64];
  __shared__ double zbuf[64];
  __shared__ double kbuf[64];
  __shared__ int ibuf[64];
  __shared__ __attribute__((aligned(16))) double colbuf[64];
  const int lane = threadIdx.x;
  const double sigma = theta[2];
  const SEParams p = se_params(theta[0], theta[1], sigma);
  const bool rv = lane < n;
  const double tr = rv ? tn[lane] : 0.0;
  const int ir = (rv && idx_n) ? idx_n[lane] : 0;
  const double zr = rv ? (yn[lane] - mean_obs[lane]) : 0.0;
  tbuf[lane] = tr;
  zbuf[lane] = zr;
  ibuf[lane] = ir;
  __syncthreads();
  double a[NMAX];
  build_row<NMAX>(a, lane, n, tr, tbuf, p);
  if (cov != nullptr) {
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (rv && c < n) a[c] += cov[(int64_t)ibuf[c] * ldc + ir];
  }
  __syncthreads();
  double logdet;
  int info;
  sweep_inverse<NMAX>(a, lane, colbuf, logdet, info);
  if (lane == 0 && blockIdx.x == 0) info_out[0] = info;
  double alpha = 0.0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) alpha = fma(a[c], zbuf[c], alpha);
  const int j0 = blockIdx.x * chunk;
  const int j1 = min(T, j0 + chunk);
  for (int j = j0; j < j1; ++j) {
    const double tt = ts[j];
    const int it = idx_t ? idx_t[j] : 0;
    const double d = tr - tt;
    double kv = rv ? exp(p.th1 - p.h * d * d) : 0.0;
    if (rv && cov != nullptr) kv += cov[(int64_t)it * ldc + ir];
    __syncthreads();
    kbuf[lane] = kv;
    __syncthreads();
    double u = 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) u = fma(a[c], kbuf[c], u);
    const double q = wave_sum(kv * u);
    const double m = wave_sum(kv * alpha);
    if (lane == 0) {
      const double ctt = (cov != nullptr) ? cov[(int64_t)it * ldc + it] : 0.0;
      out_mean[j] = mean_pred[j] + m;
      out_var[j] = (p.eth1 + p.sig2 + ctt) - q;
    }
  }
}

// ------------------------------------------------------------------ launchers
#define MAGMA_DISPATCH_NMAX(nmax, CALL)                 \
  do {                                                  \
    if ((nmax) <= 8) { CALL(8); }                       \
    else if ((nmax) <= 16) { CALL(16); }                \
    else if ((nmax) <= 24) { CALL(24); }                \
    else if ((nmax) <= 32) { CALL(32); }                \
    else if ((nmax) <= 40) { CALL(40); }                \
    else if ((nmax) <= 48) { CALL(48); }                \
    else if ((nmax) <= 52) { CALL(52); }                \
    else if ((nmax) <= 56) { CALL(56); }                \
    else { CALL(64); }                                  \
  } while (0)

int launch_se_cov_batched(int M, const int64_t* off, const int64_t* off2, const double* t,
                          const double* theta, int tstride, double* out, hipStream_t st) {
  if (M <= 0) return 0;
  int grid = M < 65536 ? M : 65536;
  hipLaunchKernelGGL(se_cov_batched_kernel, dim3(grid), dim3(256), 0, st, M, off, off2, t,
                     theta, tstride, out);
  return check_hip(hipGetLastError(), "se_cov_batched");
}

int launch_se_inv_batched(int M, int nmax, const int64_t* off, const int64_t* off2,
                          const double* t, const double* theta, int tstride, double* out,
                          double* logdet, int* info, hipStream_t st) {
  if (M <= 0) return 0;
#define CALL(NM)                                                                            \
  hipLaunchKernelGGL(se_inv_batched_kernel<NM>, dim3(M), dim3(64), 0, st, M, off, off2, t, \
                     theta, tstride, out, logdet, info)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return check_hip(hipGetLastError(), "se_inv_batched");
}

int launch_estep_scatter(int M, int nmax, const int64_t* off, const double* t, const double* y,
                         const int* idx, const double* theta, int tstride, double* P,
                         int64_t ldp, double* w, int* info, hipStream_t st) {
  if (M <= 0) return 0;
#define CALL(NM)                                                                          \
  hipLaunchKernelGGL(estep_scatter_kernel<NM>, dim3(M), dim3(64), 0, st, M, off, t, y,   \
                     idx, theta, tstride, P, ldp, w, info)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return check_hip(hipGetLastError(), "estep_scatter");
}

template <int NM>
static int launch_indiv_eval_t(int nsel, const int* sel, const int64_t* off, const double* t,
                               const double* y, const int* idx, const double* theta, int tstride,
                               const double* mean, const double* cov, int64_t ldc, int mode,
                               int want_grad, double* f, double* g, int* info, hipStream_t st) {
  const size_t full = (size_t)NM * (NM + 2) * sizeof(double);
  const size_t dyn = (want_grad && mode == 0 && cov != nullptr) ? full : 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)indiv_eval_kernel<NM>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)full);
    attr_set = true;
  }
  hipLaunchKernelGGL(indiv_eval_kernel<NM>, dim3(nsel), dim3(64), dyn, st, nsel, sel, off, t, y,
                     idx, theta, tstride, mean, cov, ldc, mode, want_grad, f, g, info);
  return check_hip(hipGetLastError(), "indiv_eval");
}

int launch_indiv_eval(int nsel, const int* sel, int nmax, const int64_t* off, const double* t,
                      const double* y, const int* idx, const double* theta, int tstride,
                      const double* mean, const double* cov, int64_t ldc, int mode, int want_grad,
                      double* f, double* g, int* info, hipStream_t st) {
  if (nsel <= 0) return 0;
  int rc = 0;
#define CALL(NM)                                                                              \
  rc = launch_indiv_eval_t<NM>(nsel, sel, off, t, y, idx, theta, tstride, mean, cov, ldc,    \
                               mode, want_grad, f, g, info, st)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return rc;
}

int launch_pred_gp(int n, const double* tn, const double* yn, const double* mean_obs,
                   const int* idx_n, int T, const double* ts, const double* mean_pred,
                   const int* idx_t, const double* cov, int64_t ldc, const double* theta,
                   double* out_mean, double* out_var, int* info, hipStream_t st) {
  if (T <= 0) return 0;
  const int chunk = 8;
  const int grid = (T + chunk - 1) / chunk;
#define CALL(NM)                                                                            \
  hipLaunchKernelGGL(pred_gp_kernel<NM>, dim3(grid), dim3(64), 0, st, n, tn, yn, mean_obs, \
                     idx_n, T, ts, mean_pred, idx_t, cov, ldc, theta, out_mean, out_var,   \
                     info, chunk)
  MAGMA_DISPATCH_NMAX(n, CALL);
#undef CALL
  return check_hip(hipGetLastError(), "pred_gp");
}

}  // namespace magma
