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
#include "mfma_sweep.h"
#include "lbfgsb_core.h"

#include <atomic>
#include <mutex>
#include <type_traits>

namespace magma {

#ifndef MAGMA_INDIV_WAVES
#define MAGMA_INDIV_WAVES 1
#endif

constexpr double LOG_2PI = 1.8378770664093454835606594728112;

// Ordering point for LDS traffic WITHIN one wavefront (every per-individual kernel gives an
// individual to one wavefront, whose LDS region no other wavefront touches): the LDS unit executes
// a wavefront's instructions in order, so data written by one lane is there for another lane's
// later read without a hardware barrier; what is needed is that the compiler keeps the two in
// program order. A workgroup barrier here would tie the four independent wavefronts of the
// device-resident M-step's workgroups together.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 2^-e with d = m 2^e, 0.5 <= m < 1: scaling a matrix whose largest diagonal entry is d by it
// brings the diagonal into (0, 1) without a single rounding (1 for a d that cannot be scaled; the
// sweep then flags the matrix by itself).
__device__ __forceinline__ double pow2_scale(double d) {
  if (!(d > 0.0) || !(d < 1.7e308)) return 1.0;
  int e;
  (void)frexp(d, &e);
  return ldexp(1.0, -e);
}

// Diagnostic build only (-DMAGMA_STAMPS): cycle shares of the segments of indiv_eval.
#ifdef MAGMA_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP_INIT() unsigned long long st_prev = stamp_now(); unsigned long long st_acc[10] = {0,0,0,0,0,0,0,0,0,0}
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long st_t = stamp_now(); st_acc[i] += st_t - st_prev; st_prev = st_t; __builtin_amdgcn_sched_barrier(0);} while (0)
#define STAMP_FLUSH() do { if (threadIdx.x == 0) { for (int q = 0; q < 10; ++q) atomicAdd(&g_stamps[q], st_acc[q]); atomicAdd(&g_stamps[15], 1ull);} } while (0)
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// inside eval_factor (slots 10 = build, 11 = sweep, 12 = K^-1 store)
#define FSTAMP_INIT() unsigned long long fs_prev = stamp_now()
#define FSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long fs_t = stamp_now(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_stamps[i], fs_t - fs_prev); fs_prev = fs_t; __builtin_amdgcn_sched_barrier(0);} while (0)
#else
#define STAMP_INIT()
#define STAMP(i)
#define STAMP_FLUSH()
#define FSTAMP_INIT()
#define FSTAMP(i)
#endif

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

// ------------------------------------------------------------------ slow path: solve() / ginv()
// `tryCatch(solve(mat), error = function(e) MASS::ginv(mat))` (CF:133,161; MG:173) for the K_i the
// Cholesky-type sweep flagged (info != 0): numerically singular or indefinite by rounding
// (sigma -> 0, near-duplicate timestamps, a huge length scale). One wavefront per flagged
// matrix, everything in LDS, plain FP64 (a slow path: correctness over speed):
//  1. solve(): Gauss-Jordan with partial pivoting on [K | I] (the pivoting of LAPACK dgesv, which
//     R's solve() calls); R raises an error when a pivot is exactly zero or the reciprocal
//     condition number is below .Machine$double.eps -- taken here as 1 / (|K|_1 |K^-1|_1) from
//     the computed inverse (R: dgecon's estimate of the same quantity).
//  2. ginv(): K is symmetric, so its SVD is its eigen-decomposition up to signs: cyclic Jacobi
//     (round-robin pairs, two-sided rotations) to K = Q diag(lambda) Q', then
//     ginv = sum over |lambda_k| > sqrt(eps) max|lambda| of q_k q_k' / lambda_k (MASS::ginv's
//     tolerance on the singular values |lambda_k|).
// Parity with R is UNPINNED for this path (no reference fixture reaches it).
// Output: K_i^-1 (or the pseudo-inverse) into Kinv_out / v_out (gather route), or added into
// P / w with atomics (P != null). logdet_out (optional): log|det K| of the LU, NaN after ginv().
// route_out[i]: 1 = solve() answered, 2 = ginv().
constexpr int SP_LD = 65;
constexpr int SP_LDS_DOUBLES = 2 * 64 * SP_LD + 8 * 64;

struct SlowInv {
  double* R;      // the answer: solve()'s inverse or ginv()'s pseudo-inverse
  double* F;      // the other 64 x SP_LD matrix, free for the caller
  int route;      // 1 = solve() answered, 2 = ginv()
  double logdet;  // log|det| of the LU (route 1), NaN after ginv()
};

// build(): lane r writes row r of the matrix into A and returns sum_c |A[r][c]| (= the 1-norm of
// column r by symmetry). Ends with a workgroup barrier; the workgroup is one wavefront.
template <class Build>
__device__ __forceinline__ SlowInv slow_solve_or_ginv(int lane, int n, double* A, double* B,
                                                      double* cbuf, double* sbuf, double* lam,
                                                      int* pq, Build&& build) {
  const double eps = 2.220446049250313e-16;
  double n1 = build();
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) n1 = fmax(n1, __shfl_xor(n1, s));
  if (lane < n)
    for (int c = 0; c < n; ++c) B[lane * SP_LD + c] = (c == lane) ? 1.0 : 0.0;
  __syncthreads();
  // ---- 1. solve(): Gauss-Jordan, partial pivoting
  bool singular = false;
  double logdet = 0.0;
  for (int k = 0; k < n; ++k) {
    double best = (lane >= k && lane < n) ? fabs(A[lane * SP_LD + k]) : -1.0;
    int bi = lane;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const double ob = __shfl_xor(best, s);
      const int oi = __shfl_xor(bi, s);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (!(best > 0.0) || !(best < 1.7e308)) {
      singular = true;
      break;
    }
    if (lane < n && bi != k) {  // swap rows k and bi (lane = column)
      const double a = A[k * SP_LD + lane], b = A[bi * SP_LD + lane];
      A[k * SP_LD + lane] = b;
      A[bi * SP_LD + lane] = a;
      const double c = B[k * SP_LD + lane], d = B[bi * SP_LD + lane];
      B[k * SP_LD + lane] = d;
      B[bi * SP_LD + lane] = c;
    }
    __syncthreads();
    const double piv = A[k * SP_LD + k];
    logdet += log(fabs(piv));
    if (lane < n) {
      cbuf[lane] = A[k * SP_LD + lane] / piv;
      sbuf[lane] = B[k * SP_LD + lane] / piv;
    }
    __syncthreads();
    if (lane < n) {
      if (lane == k) {
        for (int c = 0; c < n; ++c) {
          A[k * SP_LD + c] = cbuf[c];
          B[k * SP_LD + c] = sbuf[c];
        }
      } else {
        const double f = A[lane * SP_LD + k];
        for (int c = 0; c < n; ++c) {
          A[lane * SP_LD + c] = fma(-f, cbuf[c], A[lane * SP_LD + c]);
          B[lane * SP_LD + c] = fma(-f, sbuf[c], B[lane * SP_LD + c]);
        }
      }
    }
    __syncthreads();
  }
  int route = 1;
  if (!singular) {
    double ni = 0.0;  // 1-norm of column `lane` of the computed inverse
    bool fin = true;
    if (lane < n)
      for (int r = 0; r < n; ++r) {
        const double v = B[r * SP_LD + lane];
        ni += fabs(v);
        fin = fin && ((v - v) == 0.0);
      }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) ni = fmax(ni, __shfl_xor(ni, s));
    const bool allfin = __all(fin);
    if (!allfin || !(1.0 / (n1 * ni) >= eps)) singular = true;
  }
  double* R = B;  // where the answer is
  if (singular) {
    // ---- 2. ginv(): Jacobi eigen-decomposition of K (A), eigenvectors in B
    route = 2;
    logdet = __longlong_as_double(0x7ff8000000000000LL);
    __syncthreads();
    (void)build();
    if (lane < n)
      for (int c = 0; c < n; ++c) B[lane * SP_LD + c] = (c == lane) ? 1.0 : 0.0;
    __syncthreads();
    const int ne = n + (n & 1);  // even number of players; index n (if any) is a bye
    const int half = ne / 2;
    for (int sweep = 0; sweep < 40; ++sweep) {
      double offd = 0.0, diag = 0.0;
      if (lane < n)
        for (int c = 0; c < n; ++c) {
          const double v = A[lane * SP_LD + c];
          if (c == lane) diag += v * v; else offd += v * v;
        }
      offd = wave_sum(offd);
      diag = wave_sum(diag);
      if (!(offd > 1e-30 * diag)) break;  // wave-uniform
      for (int round = 0; round < ne - 1; ++round) {
        // round-robin tournament: player ne-1 fixed, the others rotate
        if (lane < half) {
          int a = (lane == 0) ? ne - 1 : (round + lane) % (ne - 1);
          int b = (round + ne - 1 - lane) % (ne - 1);
          if (a > b) { const int x = a; a = b; b = x; }
          double c = 1.0, sn = 0.0;
          if (b < n) {
            const double apq = A[a * SP_LD + b];
            if (apq != 0.0) {
              const double tau = (A[b * SP_LD + b] - A[a * SP_LD + a]) / (2.0 * apq);
              const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
              c = 1.0 / sqrt(1.0 + tt * tt);
              sn = tt * c;
            }
          } else {
            a = b = 0;  // bye: identity rotation on (0, 0), skipped below
          }
          pq[2 * lane] = a;
          pq[2 * lane + 1] = b;
          cbuf[lane] = c;
          sbuf[lane] = sn;
        }
        __syncthreads();
        if (lane < n) {  // columns: A <- A J, V <- V J (lane = row)
          for (int k = 0; k < half; ++k) {
            const int a = pq[2 * k], b = pq[2 * k + 1];
            if (a == b) continue;
            const double c = cbuf[k], sn = sbuf[k];
            const double x = A[lane * SP_LD + a], z = A[lane * SP_LD + b];
            A[lane * SP_LD + a] = c * x - sn * z;
            A[lane * SP_LD + b] = sn * x + c * z;
            const double u = B[lane * SP_LD + a], v = B[lane * SP_LD + b];
            B[lane * SP_LD + a] = c * u - sn * v;
            B[lane * SP_LD + b] = sn * u + c * v;
          }
        }
        __syncthreads();
        if (lane < n) {  // rows: A <- J' A (lane = column)
          for (int k = 0; k < half; ++k) {
            const int a = pq[2 * k], b = pq[2 * k + 1];
            if (a == b) continue;
            const double c = cbuf[k], sn = sbuf[k];
            const double x = A[a * SP_LD + lane], z = A[b * SP_LD + lane];
            A[a * SP_LD + lane] = c * x - sn * z;
            A[b * SP_LD + lane] = sn * x + c * z;
          }
        }
        __syncthreads();
      }
    }
    double lk = lane < n ? A[lane * SP_LD + lane] : 0.0;
    double lmax = fabs(lk);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, s));
    const double tol = 1.4901161193847656e-08 * lmax;  // sqrt(.Machine$double.eps) * d[1]
    lam[lane] = (lane < n && fabs(lk) > tol) ? 1.0 / lk : 0.0;
    __syncthreads();
    if (lane < n)  // pseudo-inverse row `lane` into A (only its own row is overwritten)
      for (int c = 0; c < n; ++c) {
        double sacc = 0.0;
        for (int k = 0; k < n; ++k) sacc = fma(B[lane * SP_LD + k] * lam[k], B[c * SP_LD + k], sacc);
        A[lane * SP_LD + c] = sacc;
      }
    R = A;
  }
  __syncthreads();
  SlowInv r;
  r.R = R;
  r.F = (R == A) ? B : A;
  r.route = route;
  r.logdet = logdet;
  return r;
}

// log|det R| through an LU with partial pivoting (R's determinant(): dgetrf) of a copy of R in W.
__device__ __forceinline__ double slow_logabsdet(int lane, int n, const double* R, double* W) {
  if (lane < n)
    for (int c = 0; c < n; ++c) W[lane * SP_LD + c] = R[lane * SP_LD + c];
  __syncthreads();
  double acc = 0.0;
  for (int k = 0; k < n; ++k) {
    double best = (lane >= k && lane < n) ? fabs(W[lane * SP_LD + k]) : -1.0;
    int bi = lane;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
      const double ob = __shfl_xor(best, s);
      const int oi = __shfl_xor(bi, s);
      if (ob > best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (!(best > 0.0)) return -__builtin_inf();                              // exactly singular
    if (!(best < 1.7e308)) return __longlong_as_double(0x7ff8000000000000LL);  // not finite
    if (lane < n && bi != k) {  // swap rows k and bi (lane = column)
      const double a = W[k * SP_LD + lane], b = W[bi * SP_LD + lane];
      W[k * SP_LD + lane] = b;
      W[bi * SP_LD + lane] = a;
    }
    __syncthreads();
    const double piv = W[k * SP_LD + k];
    acc += log(fabs(piv));
    if (lane > k && lane < n) {
      const double f = W[lane * SP_LD + k] / piv;
      for (int c = k + 1; c < n; ++c) W[lane * SP_LD + c] = fma(-f, W[k * SP_LD + c], W[lane * SP_LD + c]);
    }
    __syncthreads();
  }
  return acc;
}

__global__ __launch_bounds__(64) void small_slowpath_kernel(
    int M, const int64_t* __restrict__ off, const int64_t* __restrict__ off2,
    const double* __restrict__ t, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, const int* __restrict__ info,
    double* __restrict__ Kinv_out, double* __restrict__ v_out, double* __restrict__ logdet_out,
    double* __restrict__ P, int64_t ldp, double* __restrict__ w, int* __restrict__ route_out) {
  extern __shared__ __attribute__((aligned(16))) double sp[];
  double* A = sp;                  // [64][SP_LD]
  double* B = sp + 64 * SP_LD;     // [64][SP_LD]
  double* tbuf = B + 64 * SP_LD;   // timestamps
  double* cbuf = tbuf + 64;        // rotation cosines / row k of A
  double* sbuf = cbuf + 64;        // rotation sines / row k of B
  double* lam = sbuf + 64;         // eigenvalues
  int* pq = reinterpret_cast<int*>(lam + 64);  // [64]: pairs of the current round (p, q)
  const int lane = threadIdx.x;
  for (int i = blockIdx.x; i < M; i += gridDim.x) {
    if (info[i] == 0) continue;  // wave-uniform
    const int64_t o = off[i];
    const int n = (int)(off[i + 1] - o);
    const double* th = theta + (int64_t)i * tstride;
    const SEParams p = se_params(th[0], th[1], th[2]);
    __syncthreads();
    tbuf[lane] = lane < n ? t[o + lane] : 0.0;
    __syncthreads();
    const SlowInv sv = slow_solve_or_ginv(lane, n, A, B, cbuf, sbuf, lam, pq, [&]() {
      double colsum = 0.0;
      if (lane < n)
        for (int c = 0; c < n; ++c) {
          const double d = tbuf[lane] - tbuf[c];
          const double v = (c == lane) ? (p.eth1 + p.sig2) : exp(p.th1 - p.h * d * d);
          A[lane * SP_LD + c] = v;
          colsum += fabs(v);
        }
      return colsum;  // = 1-norm of column `lane` by symmetry
    });
    const double* R = sv.R;
    const int route = sv.route;
    const double logdet = sv.logdet;
    // ---- output (R symmetric up to rounding: row `lane` read as column `lane`)
    if (lane < n) {
      double vi = 0.0;
      if (y != nullptr)
        for (int c = 0; c < n; ++c) vi = fma(R[lane * SP_LD + c], y[o + c], vi);
      if (P != nullptr) {
        const int ir = idx[o + lane];
        unsafeAtomicAdd(&w[ir], vi);
        for (int c = 0; c <= lane; ++c)
          unsafeAtomicAdd(&P[(int64_t)idx[o + c] * ldp + ir], R[lane * SP_LD + c]);
      } else {
        if (v_out != nullptr) v_out[o + lane] = vi;
        double* Ko = Kinv_out + off2[i];
        // lower triangle from row `lane`, mirrored: the stored matrix is exactly symmetric
        for (int c = 0; c <= lane; ++c) {
          const double v = R[lane * SP_LD + c];
          Ko[(int64_t)c * n + lane] = v;
          Ko[(int64_t)lane * n + c] = v;
        }
      }
    }
    if (lane == 0) {
      if (logdet_out != nullptr) logdet_out[i] = logdet;
      if (route_out != nullptr) route_out[i] = route;
    }
  }
}

// ------------------------------------------------------------------ slow path of the evaluations
// logL_GP_mod / gr_GP_mod (CF:181-201, 304-332) and logL_GP / gr_GP (CF:174-179, 279-302) for the
// slots whose matrix the sweep of indiv_eval_kernel flagged: the reference does not stop there, it
// evaluates with `tryCatch(solve(.), error = ginv(.))` (CF:133, 177, 285) and takes the
// log-determinant from the LU of that inverse (dmvnorm, CF:27: `- determinant(inv)$modulus`).
// Same formulas, term by term, from R = solve() / ginv() of K (mode 0) or K + C (mode 1):
//   f  = (n log 2pi - log|det R| + z' R z) / 2 + [mode 0] sum(R o C) / 2
//   W  = (R z)(R z)' - R + [mode 0] R C R;  g = -(tr(W K_1) / 2, tr(W K_2) / 2, sigma tr W)
// One wavefront per flagged slot; workgroup b looks at slots 64 b .. 64 b + 63 (one load) and
// leaves at once when none is flagged. info stays as the sweep set it (nonzero = answered here).
// Parity with R is UNPINNED for this path (no reference fixture reaches it); checked against the
// oracle's restatement of the same formulas.
// C[r][c] in the packed layout of indiv_gather_cov_kernel
__device__ __forceinline__ double cpack_el(const double* __restrict__ Cq, int nt, int r, int c) {
  return Cq[(int64_t)(((c >> 2) * 64 + ((c & 3) << 4) + (r & 15)) * nt + (r >> 4))];
}

__global__ __launch_bounds__(64) void indiv_eval_slowpath_kernel(
    int nsel, const int* __restrict__ sel, const int64_t* __restrict__ off,
    const double* __restrict__ t, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, const double* __restrict__ mean,
    const double* __restrict__ cpack, int nt, int nc, int mode, int want_grad,
    double* __restrict__ f_out, double* __restrict__ g_out, const int* __restrict__ info) {
  const int lane = threadIdx.x;
  const int base = (int)blockIdx.x * 64;
  const int mine = (base + lane < nsel) ? info[base + lane] : 0;
  unsigned long long mask = __ballot(mine != 0);
  if (mask == 0ull) return;
  extern __shared__ __attribute__((aligned(16))) double sp[];
  double* A = sp;
  double* B = sp + 64 * SP_LD;
  double* tbuf = B + 64 * SP_LD;
  double* cbuf = tbuf + 64;
  double* sbuf = cbuf + 64;
  double* lam = sbuf + 64;
  int* pq = reinterpret_cast<int*>(lam + 64);
  double* zbuf = lam + 128;
  double* abuf = zbuf + 64;
  while (mask != 0ull) {
    const int io = base + (__ffsll((long long)mask) - 1);
    mask &= mask - 1ull;
    const int i = sel ? sel[io] : io;
    const int64_t o = off[i];
    const int n = (int)(off[i + 1] - o);
    const double* th = theta + (int64_t)io * tstride;
    const double sigma = th[2];
    const SEParams p = se_params(th[0], th[1], sigma);
    const double* Cq = cpack ? cpack + (int64_t)i * (16 * nt * nc) : nullptr;
    const bool with_c = (mode == 0 && Cq != nullptr);
    const bool add_c = (mode == 1 && Cq != nullptr);
    __syncthreads();
    tbuf[lane] = lane < n ? t[o + lane] : 0.0;
    zbuf[lane] = lane < n ? (y[o + lane] - mean[idx[o + lane]]) : 0.0;
    __syncthreads();
    const SlowInv sv = slow_solve_or_ginv(lane, n, A, B, cbuf, sbuf, lam, pq, [&]() {
      double colsum = 0.0;
      if (lane < n)
        for (int c = 0; c < n; ++c) {
          const double d = tbuf[lane] - tbuf[c];
          double v = (c == lane) ? (p.eth1 + p.sig2) : exp(p.th1 - p.h * d * d);
          if (add_c) v += cpack_el(Cq, nt, lane, c);
          A[lane * SP_LD + c] = v;
          colsum += fabs(v);
        }
      return colsum;
    });
    const double* R = sv.R;
    double* F = sv.F;
    const double ldr = slow_logabsdet(lane, n, R, F);
    double alpha = 0.0, rc = 0.0;
    if (lane < n)
      for (int c = 0; c < n; ++c) {
        alpha = fma(R[lane * SP_LD + c], zbuf[c], alpha);
        if (with_c) rc = fma(R[lane * SP_LD + c], cpack_el(Cq, nt, lane, c), rc);
      }
    abuf[lane] = alpha;
    const double ssq = wave_sum(zbuf[lane] * alpha);
    const double cor = wave_sum(rc);
    const double f = 0.5 * ((double)n * LOG_2PI - ldr + ssq) + 0.5 * cor;
    __syncthreads();  // abuf visible, the LU's copy in F no longer needed
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (want_grad) {
      if (with_c) {
        if (lane < n)
          for (int c = 0; c < n; ++c) {  // F = R C
            double a = 0.0;
            for (int k = 0; k < n; ++k) a = fma(R[lane * SP_LD + k], cpack_el(Cq, nt, k, c), a);
            F[lane * SP_LD + c] = a;
          }
        __syncthreads();
      }
      if (lane < n)
        for (int c = 0; c < n; ++c) {
          double wv = alpha * abuf[c] - R[lane * SP_LD + c];
          if (with_c) {
            double a = 0.0;
            for (int j = 0; j < n; ++j) a = fma(F[lane * SP_LD + j], R[j * SP_LD + c], a);
            wv += a;
          }
          const double d = tbuf[lane] - tbuf[c];
          const double gr = p.h * d * d;
          const double k1 = (c == lane) ? p.eth1 : exp(p.th1 - gr);       // CF:289: sigma = 0
          const double k2 = (c == lane) ? 0.0 : gr * exp(p.th1 - gr);     // CF:272-277, zero diagonal
          s1 = fma(wv, k1, s1);
          s2 = fma(wv, k2, s2);
          s3 += (c == lane) ? wv : 0.0;
        }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      s3 = wave_sum(s3);
    }
    if (lane == 0) {
      f_out[io] = f;
      if (want_grad) {
        g_out[3 * (int64_t)io + 0] = -0.5 * s1;
        g_out[3 * (int64_t)io + 1] = -0.5 * s2;
        g_out[3 * (int64_t)io + 2] = -sigma * s3;
      }
    }
  }
}

// pred_gp (MG:172-180) for the individuals whose cov_tn_tn the sweep flagged: inv_mat from
// solve() / ginv() (MG:173), then every target of that individual by this one wavefront.
// mean_grid != nullptr: the batched entry point's device-resident mean (gathered through idx).
__device__ __attribute__((noinline)) void pred_gp_slow_body(
    int n, const double* __restrict__ tn, const double* __restrict__ yn,
    const double* __restrict__ mean_obs, const int* __restrict__ idx_n, int T,
    const double* __restrict__ ts, const double* __restrict__ mean_pred,
    const int* __restrict__ idx_t, const double* __restrict__ mean_grid,
    const double* __restrict__ cov, int64_t ldc, const double* __restrict__ theta,
    double* __restrict__ out_mean, double* __restrict__ out_var, double* sp) {
  const int lane = threadIdx.x;
  double* A = sp;
  double* B = sp + 64 * SP_LD;
  double* tbuf = B + 64 * SP_LD;
  double* cbuf = tbuf + 64;
  double* sbuf = cbuf + 64;
  double* lam = sbuf + 64;
  int* pq = reinterpret_cast<int*>(lam + 64);
  double* zbuf = lam + 128;
  int* ibuf = reinterpret_cast<int*>(zbuf + 128);
  const SEParams p = se_params(theta[0], theta[1], theta[2]);
  const bool rv = lane < n;
  const double tr = rv ? tn[lane] : 0.0;
  const int ir = (rv && idx_n) ? idx_n[lane] : 0;
  const double mo = !rv ? 0.0 : (mean_grid ? mean_grid[ir] : mean_obs[lane]);
  __syncthreads();
  tbuf[lane] = tr;
  zbuf[lane] = rv ? (yn[lane] - mo) : 0.0;
  ibuf[lane] = ir;
  __syncthreads();
  const SlowInv sv = slow_solve_or_ginv(lane, n, A, B, cbuf, sbuf, lam, pq, [&]() {
    double colsum = 0.0;
    if (rv)
      for (int c = 0; c < n; ++c) {
        const double d = tr - tbuf[c];
        double v = (c == lane) ? (p.eth1 + p.sig2) : exp(p.th1 - p.h * d * d);
        if (cov != nullptr) v += cov[(int64_t)ibuf[c] * ldc + ir];
        A[lane * SP_LD + c] = v;
        colsum += fabs(v);
      }
    return colsum;
  });
  const double* R = sv.R;
  double alpha = 0.0;
  if (rv)
    for (int c = 0; c < n; ++c) alpha = fma(R[lane * SP_LD + c], zbuf[c], alpha);
  for (int j = 0; j < T; ++j) {
    const double tt = ts[j];
    const int it = idx_t ? idx_t[j] : 0;
    const double d = tr - tt;
    double kv = rv ? exp(p.th1 - p.h * d * d) : 0.0;
    if (rv && cov != nullptr) kv += cov[(int64_t)it * ldc + ir];
    __syncthreads();
    cbuf[lane] = kv;
    __syncthreads();
    double u = 0.0;
    if (rv)
      for (int c = 0; c < n; ++c) u = fma(R[lane * SP_LD + c], cbuf[c], u);
    const double q = wave_sum(kv * u);
    const double m = wave_sum(kv * alpha);
    if (lane == 0) {
      const double ctt = (cov != nullptr) ? cov[(int64_t)it * ldc + it] : 0.0;
      const double mp = mean_grid ? mean_grid[it] : mean_pred[j];
      out_mean[j] = mp + m;
      out_var[j] = (p.eth1 + p.sig2 + ctt) - q;
    }
  }
}

__global__ __launch_bounds__(64) void pred_gp_slowpath_kernel(
    int n, const double* __restrict__ tn, const double* __restrict__ yn,
    const double* __restrict__ mean_obs, const int* __restrict__ idx_n, int T,
    const double* __restrict__ ts, const double* __restrict__ mean_pred,
    const int* __restrict__ idx_t, const double* __restrict__ cov, int64_t ldc,
    const double* __restrict__ theta, double* __restrict__ out_mean,
    double* __restrict__ out_var, const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sp[];
  if (info[0] == 0) return;
  pred_gp_slow_body(n, tn, yn, mean_obs, idx_n, T, ts, mean_pred, idx_t, nullptr, cov, ldc, theta,
                    out_mean, out_var, sp);
}

__global__ __launch_bounds__(64) void pred_gp_batched_slowpath_kernel(
    int B, const int64_t* __restrict__ off_n, const double* __restrict__ tn,
    const double* __restrict__ yn, const int* __restrict__ idx_n,
    const int64_t* __restrict__ off_t, const double* __restrict__ ts,
    const int* __restrict__ idx_t, const double* __restrict__ mean_grid,
    const double* __restrict__ cov, int64_t ldc, const double* __restrict__ theta, int tstride,
    double* __restrict__ out_mean, double* __restrict__ out_var, const int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double sp[];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    if (info[b] == 0) continue;  // wave-uniform
    const int64_t on = off_n[b], ot = off_t[b];
    pred_gp_slow_body((int)(off_n[b + 1] - on), tn + on, yn + on, nullptr, idx_n + on,
                      (int)(off_t[b + 1] - ot), ts + ot, nullptr, idx_t + ot, mean_grid, cov, ldc,
                      theta + (int64_t)b * tstride, out_mean + ot, out_var + ot, sp);
  }
}

// ------------------------------------------------------------------ packed C_i = cov[idx_i, idx_i]
// The hyper-posterior covariance is fixed during an M-step while every individual is evaluated
// tens of times: the random gather (n^2 isolated 8-byte reads, one cache line each) is done
// once; evaluations then stream a contiguous block. Slot i holds C_i (zero padded) in the
// order the evaluation kernel's lanes consume it: [kk][lane][it] = C[16 it + l][4 kk + g]
// (l = lane & 15, g = lane >> 4; kk < NC / 4, it < NT, NC = the kernel instantiation nmax
// rounds up to, NT = ceil(NC / 16)). One wavefront-wide 16-byte load then fetches 1 KB of
// contiguous memory, and by symmetry the same address serves both as the MFMA A fragment of
// k-block kk and as rows 4 kk + g of C in accumulator layout.
__global__ __launch_bounds__(256) void indiv_gather_cov_kernel(
    int M, const int64_t* __restrict__ off, const int* __restrict__ idx,
    const double* __restrict__ cov, int64_t ldc, double* __restrict__ cpack, int nt, int nc) {
  // cov is symmetric: only the n (n + 1) / 2 elements on or below the diagonal are fetched (each
  // one an isolated 8-byte read that costs a whole memory sector) and mirrored through LDS
  constexpr int LDC = MAGMA_NSMALL_MAX + 1;
  __shared__ int ibuf[MAGMA_NSMALL_MAX];
  __shared__ double Cl[MAGMA_NSMALL_MAX * LDC];
  for (int i = blockIdx.x; i < M; i += gridDim.x) {
    const int64_t o = off[i];
    const int n = (int)(off[i + 1] - o);
    __syncthreads();
    if ((int)threadIdx.x < n) ibuf[threadIdx.x] = idx[o + threadIdx.x];
    __syncthreads();
    // all of a thread's reads are in flight before the first is used: written as a plain loop
    // (read, two LDS stores, next read) the compiler waits for every read in turn, and the
    // kernel's time is (n^2 / 256) serial memory latencies per individual
    constexpr int SLOTS = MAGMA_NSMALL_MAX * MAGMA_NSMALL_MAX / 256;
    double gv[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int e = (int)threadIdx.x + 256 * q;
      const int r = e % n, c = e / n;
      gv[q] = (e < n * n && r >= c) ? cov[(int64_t)ibuf[c] * ldc + ibuf[r]] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
      const int e = (int)threadIdx.x + 256 * q;
      const int r = e % n, c = e / n;
      if (e < n * n && r >= c) {
        Cl[r * LDC + c] = gv[q];
        Cl[c * LDC + r] = gv[q];
      }
    }
    __syncthreads();
    const int tot = 16 * nt * nc;
    double* out = cpack + (int64_t)i * tot;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
      const int it = e % nt, ln = (e / nt) & 63, kk = e / (64 * nt);
      const int r = 16 * it + (ln & 15), c = 4 * kk + (ln >> 4);
      out[e] = (r < n && c < n) ? Cl[r * LDC + c] : 0.0;
    }
  }
}

// ------------------------------------------------------------------ common timestamps + common HPs
// CF:221-224, 347-350: with identical timestamps and common hyper-parameters the reference
// inverts K once per evaluation and only the residuals differ between individuals. The sum over
// the M individuals of the objective (CF:203-234) and of its gradient (CF:334-368) is then M
// times the objective / gradient of ONE pseudo-individual with residual 0 and
// C' = C + (1/M) sum_i r_i r_i' in place of C:
//   sum_i [r_i' K^-1 r_i + tr(K^-1 C)] = M tr(K^-1 C'),
//   sum_i [a_i a_i' + K^-1 (C K^-1 - I)] = M [K^-1 C' K^-1 - K^-1],  a_i = K^-1 r_i.
// dedup_outer_kernel forms S / M (fixed summation order), dedup_pack_kernel the
// packed C' (layout of indiv_gather_cov_kernel) and the pseudo-individual's outputs y' = mean
// at its timestamps (so that its residual is exactly 0).
constexpr int DEDUP_PARTS = 16;
__global__ __launch_bounds__(64 * DEDUP_PARTS) void dedup_outer_kernel(
    int M, int n, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ mean, double* __restrict__ S) {
  // workgroup b = column b of S; thread (part, a): individuals part, part + 16, ... in order; the
  // sixteen partial sums are then added in part order (fixed summation order: repeatable)
  __shared__ double red[DEDUP_PARTS][64];
  const int b = blockIdx.x, a = threadIdx.x & 63, part = threadIdx.x >> 6;
  double acc = 0.0;
  if (a < n && b < n) {
    const double ma = mean[idx[a]], mb = mean[idx[b]];
    for (int i = part; i < M; i += DEDUP_PARTS)
      acc = fma(y[(int64_t)i * n + a] - ma, y[(int64_t)i * n + b] - mb, acc);
  }
  red[part][a] = acc;
  __syncthreads();
  if (part == 0 && a < n && b < n) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < DEDUP_PARTS; ++q) tot += red[q][a];
    S[a + (int64_t)n * b] = tot / (double)M;
  }
}

__global__ __launch_bounds__(256) void dedup_pack_kernel(
    int n, const int* __restrict__ idx, const double* __restrict__ cov, int64_t ldc,
    const double* __restrict__ S, const double* __restrict__ mean, double* __restrict__ cpack,
    double* __restrict__ y_out, int nt, int nc) {
  const int tot = 16 * nt * nc;
  for (int e = threadIdx.x; e < tot; e += blockDim.x) {
    const int it = e % nt, ln = (e / nt) & 63, kk = e / (64 * nt);
    const int r = 16 * it + (ln & 15), c = 4 * kk + (ln >> 4);
    cpack[e] = (r < n && c < n) ? cov[(int64_t)idx[c] * ldc + idx[r]] + S[r + (int64_t)n * c] : 0.0;
  }
  if ((int)threadIdx.x < n) y_out[threadIdx.x] = mean[idx[threadIdx.x]];
}

// ------------------------------------------------------------------ E-step, deterministic route
// The same sums as estep_scatter_kernel without atomics, so that an E-step is repeatable bit for
// bit: K_i^-1 and v_i = K_i^-1 y_i materialised by se_inv_batched_kernel (flagged individuals
// zeroed: they contribute nothing), then every element of P (and of w) that receives
// anything sums its sources in a fixed order from a list built once per dataset
// (csr_gather_sum_kernel: out[tgt[k]] = sum_{e in [ptr[k], ptr[k+1])} val[src[e]]).
__global__ __launch_bounds__(256) void csr_gather_sum_kernel(
    int64_t ntgt, const int64_t* __restrict__ ptr, const int* __restrict__ src,
    const int64_t* __restrict__ tgt, const double* __restrict__ val, double* __restrict__ out, int lg) {
  // Workgroups are dealt to the eight XCDs in turn, and each XCD has its own L2. Targets are listed
  // column by column of P, and the eight doubles of one line of a K_i^-1 go to one column of P, a
  // few hundred rows apart: with block b on targets 256 b .. 256 b + 255 those rows sit on
  // neighbouring workgroups = different XCDs, and every line is fetched about twice. So the list
  // is cut into groups of 8 * 2^lg blocks and XCD x takes 2^lg consecutive blocks of every group
  // (same sums, same order within each; the kernel itself is bound by the scattered loads each
  // compute unit can keep in flight, not by the bytes).
  const int nb = (int)gridDim.x, b = (int)blockIdx.x;
  int64_t blk = b;
  const int gs = 8 << lg;
  if (b < nb / gs * gs) {  // groups of 8 * 2^lg blocks: XCD x takes 2^lg consecutive blocks of each
    const int x = b & 7, j = (b >> 3) & ((1 << lg) - 1), g = b >> (3 + lg);
    blk = (int64_t)g * gs + (x << lg) + j;
  }
  const int64_t k = blk * 256 + threadIdx.x;
  if (k >= ntgt) return;
  double s = 0.0;
  for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e) s += val[src[e]];
  out[tgt[k]] = s;
}

// The same for long lists (w: a few hundred sources per grid point): one wavefront per target,
// lane l adds the sources ptr[k] + l, + 64, ... in order, then a fixed-order wave reduction.
__global__ __launch_bounds__(256) void csr_gather_sum_wave_kernel(
    int64_t ntgt, const int64_t* __restrict__ ptr, const int* __restrict__ src,
    const int64_t* __restrict__ tgt, const double* __restrict__ val, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= ntgt) return;
  double s = 0.0;
  for (int64_t e = ptr[k] + lane; e < ptr[k + 1]; e += 64) s += val[src[e]];
  s = wave_sum(s);
  if (lane == 0) out[tgt[k]] = s;
}

// ------------------------------------------------------------------ update_inv / update_mean
// CF:440-478 on explicit blocks: individual i's K_i^-1 (n_i x n_i, column-major at Kinv + off2[i])
// is added into P at [idx_i, idx_i] (whole block: P is treated as a full matrix, as the
// reference does) and K_i^-1 y_i into w at idx_i. A timestamp that is not on the grid (idx < 0)
// is dropped the way `intersect(row.names(...))` drops it (CF:451, 474): its row and column of
// K_i^-1 are not added to P; K_i^-1 y_i is still formed from the WHOLE block (CF:468) and only the
// rows that are on the grid are added to w. Any n_i; one workgroup per individual.
__global__ __launch_bounds__(256) void update_inv_kernel(
    int M, const int64_t* __restrict__ off, const int64_t* __restrict__ off2,
    const int* __restrict__ idx, const double* __restrict__ Kinv, const double* __restrict__ y,
    double* __restrict__ P, int64_t ldp, double* __restrict__ w) {
  for (int i = blockIdx.x; i < M; i += gridDim.x) {
    const int64_t o = off[i];
    const int n = (int)(off[i + 1] - o);
    const double* K = Kinv + off2[i];
    const int* ix = idx + o;
    if (P != nullptr)
      for (int64_t e = threadIdx.x; e < (int64_t)n * n; e += blockDim.x) {
        const int r = ix[(int)(e % n)], c = ix[(int)(e / n)];
        if (r >= 0 && c >= 0) unsafeAtomicAdd(&P[(int64_t)c * ldp + r], K[e]);
      }
    if (w != nullptr)
      for (int r = threadIdx.x; r < n; r += blockDim.x) {
        if (ix[r] < 0) continue;
        double s = 0.0;
        for (int c = 0; c < n; ++c) s = fma(K[(int64_t)c * n + r], y[o + c], s);
        unsafeAtomicAdd(&w[ix[r]], s);
      }
  }
}

// ------------------------------------------------------------------ M-step evaluation
// mode 0 (logL_GP_mod / gr_GP_mod): K = K(theta, sigma);
//        f = 1/2 (n log 2pi + log|K| + z'K^-1 z) + 1/2 sum(K^-1 o C),
//        W = aa' + K^-1 C K^-1 - K^-1, g = -(1/2 sum W o K_nl, 1/2 sum W o dK2, sigma tr W).
// mode 1 (logL_GP / gr_GP): K = K(theta, sigma) + C; f without the trace term; W = aa' - K^-1.
// C = packed cov[idx, idx] (cpack == nullptr: zero).
// Workgroup b handles individual sel[b] (sel == nullptr: b), so a lock-step optimiser only
// launches the problems that are still running.
//
// One wavefront per individual, every n^3 stage on v_mfma_f64_16x16x4_f64. The matrix lives
// in accumulator layout: 16 x 16 tiles, register v of lane (l, g) [l = lane & 15, g = lane >> 4]
// of tile (it, jt) = element (16 it + g + 4 v, 16 jt + l). MFMA operands: A fragment lane
// (l, g) = A[l][g], B fragment lane (l, g) = B[g][l]. The identity that removes all operand
// shuffling: rows 4 kk .. 4 kk + 3 of a matrix held in accumulator layout ARE its B fragment
// for k-block kk (register kk & 3 of the tiles in tile row kk >> 2).
//
// Everything symmetric is kept as its tiles on and below the diagonal only (round 3): K and K^-1
// in registers through the sweep (80 instead of 128 accumulator registers at NMAX = 52, 10 instead
// of 16 MFMAs per pivot block: mfma_sweep_lower) and K^-1 in LDS (KL below: 16 KB instead of
// 21.6 KB). Together with T = C K^-1 held whole (so that the packed C is streamed through eight
// registers instead of sitting in 104) an evaluation fits 256 registers and 20 KB of LDS, and two
// wavefronts share a SIMD: one's pivot chain, exp()s and LDS round trips run under the other's
// MFMAs.
//
//  1. K built in place (identity padded to NMAX, a multiple of 4), lower tiles.
//  2. Inverse + log-det by the blocked symmetric sweep (mfma_sweep.h): block J = 4 pivots,
//     P = A[J,J], D = P^-1 (4 x 4, every lane redundantly, LDL'), A <- A - F' B' - 2 I_J.
//     After NMAX / 4 blocks A = -K^-1.
//  3. K^-1 -> LDS once (Ks, layout KL).
//  4. T = C K^-1: A fragments straight from the packed C in global memory, B from Ks.
//  5. Z = K^-1 T: A fragments from Ks, B fragments = T's registers; only tiles on or below the
//     diagonal, each contracted with K_nl and dK/dtheta2 as soon as it is complete.
// The evaluation comes in pieces so that the device-resident M-step can hand the column blocks of
// stages 4 / 5 to other wavefronts of its workgroup (indiv_mstep_kernel); indiv_eval_body below
// strings them together for one wavefront.

// K^-1 (or K on its way in) of one individual in LDS: the 16 x 16 tiles (it, jt), jt <= it, one after
// the other by tile row, each tile row-major with a row stride of TS = 18 doubles. A fragment read
// takes 4 rows x 16 columns of a tile (lanes: l along the row, g over the rows) or, for a tile
// above the diagonal, 16 rows x 4 columns of its mirror image (l over the rows, g along them): with
// a stride of 18 doubles = 36 banks the 32 lanes of a ds_read_b64 phase fall on 32 different bank
// pairs in the second form and on all but two in the first.
template <int NMAX>
struct KL {
  static constexpr int NT = (NMAX + 15) / 16;
  static constexpr int TS = 18;
  static constexpr int rows(int it) { return NMAX - 16 * it >= 16 ? 16 : NMAX - 16 * it; }
  static constexpr int tb(int it, int jt) {  // first double of tile (it, jt), jt <= it
    int s = 0;
    for (int a = 0; a < it; ++a) s += (a + 1) * rows(a) * TS;
    return s + jt * rows(it) * TS;
  }
  static constexpr int total = tb(NT - 1, NT - 1) + rows(NT - 1) * TS;
  // (reads of a short last tile row in the second form run up to 15 * TS + 16 doubles past a tile:
  // what they fetch only reaches masked results; callers keep that much LDS behind Ks)
  static constexpr int alloc = total + 16;
  static_assert(total >= 64 * NT, "the head of Ks doubles as the sweep's transposition buffer");
};

// K^-1[4 kk + g][16 jt + l] = K^-1[16 jt + l][4 kk + g]: the B fragment of k-block kk for tile
// column jt, and the A fragment of tile row jt for k-block kk (kk, jt compile-time after unrolling).
template <int NMAX>
__device__ __forceinline__ double kinv_frag(const double* Ks, int kk, int jt, int l, int g) {
  using L = KL<NMAX>;
  const int kt = kk >> 2, kr = 4 * (kk & 3);
  if (kt >= jt) return Ks[L::tb(kt, jt) + (kr + g) * L::TS + l];
  return Ks[L::tb(jt, kt) + l * L::TS + kr + g];
}

// Stages 1-3: K(theta) (+ C in mode 1) -> K^-1 in Ks (LDS), log|K|, info. Ends with a wave_sync().
// FROM_LDS: K (unscaled, zero beyond n) was put into Ks by eval_build_row, by this wavefront or by
// others; stage 1 then only loads it.
template <int NMAX, bool FROM_LDS = false>
__device__ __forceinline__ void eval_factor(int lane, int n, int i, const SEParams& p,
                                            const double* tbuf, double* Ks,
                                            const double* __restrict__ cpack, int mode,
                                            double& logdet, int& info_res) {
  using L = KL<NMAX>;
  constexpr int NT = (NMAX + 15) / 16;  // 16 x 16 tiles per dimension
  constexpr int NK = NMAX / 4;          // blocks of 4 rows / k-blocks
  static_assert(NMAX % 4 == 0, "NMAX must be a multiple of 4");
  const int l = lane & 15, g = lane >> 4;
  const bool have_c = (cpack != nullptr);
  // packed C_i: Cq[kk * 64 * NT + it] = C[16 it + l][4 kk + g] (see indiv_gather_cov_kernel)
  const double* Cq = cpack + (int64_t)i * (16 * NT * NMAX) + lane * NT;

  // ---- 1. K in accumulator layout, tiles jt <= it (rows >= NMAX stay zero: compile-time skip)
  FSTAMP_INIT();
  v4d A[NT][NT];
  if constexpr (FROM_LDS) {
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int jt = 0; jt <= it; ++jt)
          A[it][jt][v] = (16 * it + 4 * v < NMAX) ? Ks[L::tb(it, jt) + (4 * v + g) * L::TS + l] : 0.0;
    wave_sync();  // every load is issued before the sweep reuses the head of Ks
  } else {
    double tcol[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) tcol[jt] = tbuf[16 * jt + l];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (16 * it + 4 * v >= NMAX) {
#pragma unroll
          for (int jt = 0; jt <= it; ++jt) A[it][jt][v] = 0.0;
          continue;
        }
        const int r = 16 * it + 4 * v + g;
        const double trow = tbuf[r];
#pragma unroll
        for (int jt = 0; jt <= it; ++jt) {
          const int c = 16 * jt + l;
          const double d = trow - tcol[jt];
          double kv = exp(p.th1 - p.h * d * d);
          kv = (r == c) ? (p.eth1 + p.sig2) : kv;
          A[it][jt][v] = (r < n && c < n) ? kv : 0.0;
          __builtin_amdgcn_sched_barrier(0);  // keep the unrolled exp() bodies sequential
        }
      }
    }
  }
  if (mode == 1 && have_c) {  // K + C (C is zero in the padding)
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int jt = 0; jt <= (kk >> 2); ++jt) A[kk >> 2][jt][kk & 3] += Cq[kk * 64 * NT + jt];
  }
  // Scale to a diagonal <= 1 (the sweep's operand trick is exact to rounding only for pivots
  // <= 1: mfma_sweep.h): K' = sc K, identity in the padding; K^-1 = sc K'^-1,
  // log|K| = log|K'| - n log sc.
  double dmax = p.eth1 + p.sig2;
  if (mode == 1 && have_c) {
    double dm = 0.0;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (16 * it + 4 * v < NMAX) dm = fmax(dm, (g + 4 * v == l) ? A[it][it][v] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o));
    dmax = dm;
  }
  // (a power of two: the scaling and its undoing are exact)
  const double sc = pow2_scale(dmax);
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * it + 4 * v >= NMAX) continue;
      const int r = 16 * it + 4 * v + g;
#pragma unroll
      for (int jt = 0; jt <= it; ++jt) {
        const int c = 16 * jt + l;
        A[it][jt][v] = (r < n && c < n) ? A[it][jt][v] * sc : ((r == c) ? 1.0 : 0.0);
      }
    }

  // ---- 2. blocked symmetric sweep on the lower tiles (mfma_sweep.h)
  FSTAMP(10);
  double pm = 1.0;
  int pe = 0, info = 0;
  mfma_sweep_lower<NT, NK>(A, Ks, l, g, pm, pe, info);
  FSTAMP(11);
  logdet = log(pm) + (double)pe * 0.69314718055994530941723212145818 - (double)n * log(sc);
  {
    // R's solve() refuses a matrix whose reciprocal condition number is below eps (CF:133 then
    // takes ginv). The cheap stand-in here: the largest diagonal entry of the inverse of the
    // scaled matrix (diagonal in [0.5, 1)) bounds its condition number from below; beyond 2^30
    // the matrix goes to the slow path, which applies R's own rule (LU, rcond, ginv): see
    // mfma_sweep.h, SWEEP_PIVOT_MIN.
    double dinv = 0.0;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (16 * it + 4 * v < NMAX) dinv = fmax(dinv, (g + 4 * v == l) ? -A[it][it][v] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dinv = fmax(dinv, __shfl_xor(dinv, o));
    info = (info == 0 && !(dinv < SWEEP_INVDIAG_MAX)) ? n + 1 : info;
  }
  info_res = info;

  // ---- 3. K^-1 = -A to LDS (whole tiles: columns beyond NMAX are exact zeros)
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * it + 4 * v >= NMAX) continue;
#pragma unroll
      for (int jt = 0; jt <= it; ++jt) Ks[L::tb(it, jt) + (4 * v + g) * L::TS + l] = -sc * A[it][jt][v];
    }
  wave_sync();
  FSTAMP(12);
}

// Tile row IT of K (mode 0: no C), unscaled, as eval_factor's stage 1 computes it, into Ks.
template <int NMAX, int IT>
__device__ __forceinline__ void eval_build_row(int lane, int n, const SEParams& p,
                                               const double* tbuf, double* Ks) {
  using L = KL<NMAX>;
  const int l = lane & 15, g = lane >> 4;
  double tcol[IT + 1];
#pragma unroll
  for (int jt = 0; jt <= IT; ++jt) tcol[jt] = tbuf[16 * jt + l];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    if (16 * IT + 4 * v >= NMAX) continue;
    const int r = 16 * IT + 4 * v + g;
    const double trow = tbuf[r];
#pragma unroll
    for (int jt = 0; jt <= IT; ++jt) {
      const int c = 16 * jt + l;
      const double d = trow - tcol[jt];
      double kv = exp(p.th1 - p.h * d * d);
      kv = (r == c) ? (p.eth1 + p.sig2) : kv;
      Ks[L::tb(IT, jt) + (4 * v + g) * L::TS + l] = (r < n && c < n) ? kv : 0.0;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The packed C_i as one batch of 16-byte loads (26 per lane at NMAX = 52): for the wavefronts
// that work on single column blocks (helped evaluations). (Left alone the compiler issues them a
// pair at a time next to their uses: NK serial HBM latencies per wavefront; callers put a
// sched_barrier on either side.)
template <int NMAX>
__device__ __forceinline__ void eval_load_c(double (&cf)[NMAX / 4][(NMAX + 15) / 16],
                                            const double* __restrict__ cpack, int i, int lane) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  const double* Cq = cpack + (int64_t)i * (16 * NT * NMAX) + lane * NT;
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int it = 0; it < NT; ++it) cf[kk][it] = Cq[kk * 64 * NT + it];
}

// alpha = K^-1 z (lane r owns row r: its part left of / on the diagonal tile read along the row,
// the rest down column r of the tile rows below); zero beyond NMAX.
template <int NMAX>
__device__ __forceinline__ double eval_alpha(int lane, const double* Ks, const double* zbuf) {
  using L = KL<NMAX>;
  constexpr int NT = L::NT, TS = L::TS;
  const int lr = lane < NMAX ? lane : 0;
  const int rt = lr >> 4, rl = lr & 15;
  const int rsz = ((rt == NT - 1) ? L::rows(NT - 1) : 16) * TS;  // one tile of this lane's tile row
  const int base_r = 16 * TS * (rt * (rt + 1) / 2) + rl * TS;    // (every tile row above is full)
  double alpha = 0.0;
#pragma unroll
  for (int c = 0; c < NMAX; c += 2) {
    const double2 zz = *reinterpret_cast<const double2*>(&zbuf[c]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ct = (c + q) >> 4, cl = (c + q) & 15;
      const int a_row = base_r + ct * rsz + cl;
      const int a_col = L::tb(ct, 0) + rt * (L::rows(ct) * TS) + cl * TS + rl;
      alpha = fma(Ks[rt >= ct ? a_row : a_col], q == 0 ? zz.x : zz.y, alpha);
    }
  }
  return lane < NMAX ? alpha : 0.0;
}

// Stage 4 for column block JT from a whole copy of the packed C: T[:, JT] = (C K^-1)[:, JT].
template <int NMAX, int JT>
__device__ __forceinline__ void eval_tblock(int lane, const double (&cf)[NMAX / 4][(NMAX + 15) / 16],
                                            const double* Ks, v4d (&T)[(NMAX + 15) / 16][(NMAX + 15) / 16]) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  const int l = lane & 15, g = lane >> 4;
#pragma unroll
  for (int it = 0; it < NT; ++it) T[it][JT] = (v4d){0.0, 0.0, 0.0, 0.0};
  double bf = kinv_frag<NMAX>(Ks, 0, JT, l, g);
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    const double b0 = bf;
    if (kk + 1 < NK) bf = kinv_frag<NMAX>(Ks, kk + 1, JT, l, g);
#pragma unroll
    for (int it = 0; it < NT; ++it)
      T[it][JT] = __builtin_amdgcn_mfma_f64_16x16x4f64(cf[kk][it], b0, T[it][JT], 0, 0, 0);
  }
}

// Stage 4 for all column blocks at once by one wavefront: the packed C is streamed through a ring
// of CDEPTH k-blocks' worth of registers (each k-block feeds 16 MFMAs = 1024 matrix-core cycles;
// a 16-byte load that misses L2 takes a few thousand cycles with every compute unit streaming, so
// the loads run CDEPTH k-blocks ahead; eval_cstream_start issues the first ones before alpha is
// formed). Every T[it][jt] accumulates its k-blocks in the order eval_tblock uses: same bits.
constexpr int CDEPTH = 4;
template <int NMAX>
struct CStream {
  double ring[CDEPTH][(NMAX + 15) / 16];
};
template <int NMAX>
__device__ __forceinline__ void eval_cstream_start(CStream<NMAX>& cs, int lane, int i,
                                                   const double* __restrict__ cpack) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  const double* Cq = cpack + (int64_t)i * (16 * NT * NMAX) + lane * NT;
#pragma unroll
  for (int d = 0; d < CDEPTH; ++d)
#pragma unroll
    for (int it = 0; it < NT; ++it) cs.ring[d][it] = (d < NK) ? Cq[d * 64 * NT + it] : 0.0;
}
template <int NMAX>
__device__ __forceinline__ void eval_tfull(int lane, int i, const double* __restrict__ cpack,
                                           CStream<NMAX>& cs, const double* Ks,
                                           v4d (&T)[(NMAX + 15) / 16][(NMAX + 15) / 16]) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  const int l = lane & 15, g = lane >> 4;
  const double* Cq = cpack + (int64_t)i * (16 * NT * NMAX) + lane * NT;
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) T[it][jt] = (v4d){0.0, 0.0, 0.0, 0.0};
  double bn[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) bn[jt] = kinv_frag<NMAX>(Ks, 0, jt, l, g);
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) {
    double c0[NT], b0[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      c0[it] = cs.ring[kk % CDEPTH][it];
      b0[it] = bn[it];
    }
    if (kk + CDEPTH < NK) {
#pragma unroll
      for (int it = 0; it < NT; ++it) cs.ring[kk % CDEPTH][it] = Cq[(kk + CDEPTH) * 64 * NT + it];
    }
    if (kk + 1 < NK) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) bn[jt] = kinv_frag<NMAX>(Ks, kk + 1, jt, l, g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int it = 0; it < NT; ++it)
        T[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[it], b0[jt], T[it][jt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Stage 5 for column block JT, given T[:, JT]: Z[at][JT] = K^-1 T[:, JT] for at >= JT, contracted at
// once with K_nl and dK/dtheta2; tr(C K^-1)'s share is T's diagonal. Returns THIS block's sums per
// lane, started from zero: callers reduce them over the wavefront and add the blocks in the order
// 0, 1, ..., whoever computed them, so that every route gives the same bits. alpha (abuf) is first
// read after the products: wait_alpha() is called in front of that read (a helper wavefront waits
// there for the owner).
template <int NMAX, int JT, class WaitAlpha>
__device__ __forceinline__ void eval_zblock(int lane, int n, const SEParams& p, bool with_c,
                                            const v4d (&T)[(NMAX + 15) / 16][(NMAX + 15) / 16],
                                            const double* Ks, const double* tbuf, const double* abuf,
                                            WaitAlpha&& wait_alpha, double& s1, double& s2,
                                            double& s3, double& cor) {
  using L = KL<NMAX>;
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  constexpr int jt = JT;
  const int l = lane & 15, g = lane >> 4;
  s1 = 0.0;
  s2 = 0.0;
  s3 = 0.0;
  cor = 0.0;
  v4d Z[NT];
#pragma unroll
  for (int it = 0; it < NT; ++it) Z[it] = (v4d){0.0, 0.0, 0.0, 0.0};
  if (with_c) {
    // tr(C K^-1): the diagonal of T sits in the diagonal tiles where g + 4v == l
#pragma unroll
    for (int v = 0; v < 4; ++v)
      if (16 * jt + 4 * v < NMAX) cor += (g + 4 * v == l) ? T[jt][jt][v] : 0.0;
    double ka[NT];
#pragma unroll
    for (int at = jt; at < NT; ++at) ka[at] = kinv_frag<NMAX>(Ks, 0, at, l, g);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      double k0[NT];
#pragma unroll
      for (int at = jt; at < NT; ++at) {
        k0[at] = ka[at];
        if (kk + 1 < NK) ka[at] = kinv_frag<NMAX>(Ks, kk + 1, at, l, g);
      }
#pragma unroll
      for (int at = jt; at < NT; ++at)
        Z[at] = __builtin_amdgcn_mfma_f64_16x16x4f64(k0[at], T[kk >> 2][jt][kk & 3], Z[at], 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  wait_alpha();
  // element (ra, cj), ra = 16 at + 4v + g, cj = 16 jt + l
  const int cj = 16 * jt + l;
  const double acj = abuf[cj];
  const double tcj = tbuf[cj];
#pragma unroll
  for (int at = jt; at < NT; ++at) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * at + 4 * v >= NMAX) continue;
      const int ra = 16 * at + 4 * v + g;
      const bool ok = ra < n && cj < n;
      const double d = tbuf[ra] - tcj;
      const double gq = p.h * d * d;
      double knl = exp(p.th1 - gq);
      knl = (ra == cj) ? p.eth1 : knl;
      const double W = fma(abuf[ra], acj, Z[at][v] - Ks[L::tb(at, jt) + (4 * v + g) * L::TS + l]);
      const double wt = (jt == at) ? 1.0 : 2.0;  // tiles above the diagonal by symmetry
      const double wk = ok ? wt * W * knl : 0.0;
      s1 += wk;
      s2 = fma(wk, gq, s2);  // dK/dtheta2 = gq * K_nl, zero on the diagonal (gq = 0 there)
      s3 += (ok && ra == cj) ? W : 0.0;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// f(std::integral_constant<int, jt>) for jt = 0 .. NT - 1, in order.
template <int NMAX, int JT = 0, class F>
__device__ __forceinline__ void for_blocks(F&& f) {
  f(std::integral_constant<int, JT>{});
  if constexpr (JT + 1 < (NMAX + 15) / 16) for_blocks<NMAX, JT + 1>(f);
}

// A column block's sums over the wavefront (every lane gets the same): what the blocks' owners and
// helpers exchange, four doubles per block.
struct BlockSums {
  double s1, s2, s3, cor;
};
__device__ __forceinline__ BlockSums block_reduce(double q1, double q2, double q3, double qc) {
  BlockSums b;
  b.s1 = wave_sum(q1);
  b.s2 = wave_sum(q2);
  b.s3 = wave_sum(q3);
  b.cor = wave_sum(qc);
  return b;
}

// f and the gradient from the sums of all column blocks (every lane returns the same).
__device__ __forceinline__ void eval_finish(int n, double sigma, double logdet, double ssq,
                                            int info, double s1, double s2, double s3,
                                            double cor, double& f_res, double (&g_res)[3]) {
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  f_res = (info == 0) ? 0.5 * ((double)n * LOG_2PI + logdet + ssq) + 0.5 * cor : nanv;
  g_res[0] = (info == 0) ? -0.5 * s1 : nanv;
  g_res[1] = (info == 0) ? -0.5 * s2 : nanv;
  g_res[2] = (info == 0) ? -sigma * s3 : nanv;
}

// Stages 4 / 5 by one wavefront on its own. Returns the sums over all column blocks.
template <int NMAX>
__device__ __forceinline__ BlockSums eval_products(int lane, int n, int i, const SEParams& p,
                                                   bool with_c, const double* __restrict__ cpack,
                                                   CStream<NMAX>& cs, const double* Ks,
                                                   const double* tbuf, const double* abuf) {
  constexpr int NT = (NMAX + 15) / 16;
  v4d T[NT][NT];
  if (with_c) eval_tfull<NMAX>(lane, i, cpack, cs, Ks, T);
  BlockSums tot = {0.0, 0.0, 0.0, 0.0};
  for_blocks<NMAX>([&](auto JT) {
    double q1, q2, q3, qc;
    eval_zblock<NMAX, decltype(JT)::value>(lane, n, p, with_c, T, Ks, tbuf, abuf, [] {}, q1, q2, q3, qc);
    const BlockSums b = block_reduce(q1, q2, q3, qc);
    tot.s1 += b.s1;
    tot.s2 += b.s2;
    tot.s3 += b.s3;
    tot.cor += b.cor;
  });
  return tot;
}

// One evaluation at hyper-parameters (th1, th2, sigma) by this wavefront. tbuf / zbuf hold the
// individual's timestamps and residuals y - mean (zero beyond n), abuf and Ks are scratch; every
// lane returns the same f_res, g_res (only with want_grad), info_res.
template <int NMAX>
__device__ __forceinline__ void indiv_eval_body(
    int lane, int n, int i, double th1, double th2, double sigma, const double* tbuf,
    const double* zbuf, double* abuf, double* Ks, const double* __restrict__ cpack, int mode,
    int want_grad, double& f_res, double (&g_res)[3], int& info_res) {
  using L = KL<NMAX>;
  constexpr int NT = (NMAX + 15) / 16;
  constexpr int NK = NMAX / 4;
  const int l = lane & 15, g = lane >> 4;
  const SEParams p = se_params(th1, th2, sigma);
  const double zr = zbuf[lane];
  STAMP_INIT();
  const bool with_c = (mode == 0 && cpack != nullptr);
  double logdet;
  int info;
  eval_factor<NMAX>(lane, n, i, p, tbuf, Ks, cpack, mode, logdet, info);
  STAMP(1);
  // the first k-blocks of the packed C: in flight while alpha is formed
  __builtin_amdgcn_sched_barrier(0);
  CStream<NMAX> cs;
  if (with_c && want_grad) eval_cstream_start<NMAX>(cs, lane, i, cpack);
  __builtin_amdgcn_sched_barrier(0);
  const double alpha = eval_alpha<NMAX>(lane, Ks, zbuf);
  const double ssq = wave_sum(zr * alpha);
  STAMP(2);

  if (!want_grad) {
    const double nanv = __longlong_as_double(0x7ff8000000000000LL);
    double cor = 0.0;
    if (with_c) {  // sum(K^-1 o C) over the lower tiles (twice those below the diagonal)
      const double* Cq = cpack + (int64_t)i * (16 * NT * NMAX) + lane * NT;
      double cl[NK][NT];  // one batch of loads (left alone: one memory latency per k-block)
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int jt = 0; jt <= (kk >> 2); ++jt) cl[kk][jt] = Cq[kk * 64 * NT + jt];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
#pragma unroll
        for (int jt = 0; jt <= (kk >> 2); ++jt) {
          const double kv = Ks[L::tb(kk >> 2, jt) + (4 * (kk & 3) + g) * L::TS + l];
          cor = fma(((kk >> 2) == jt ? 1.0 : 2.0) * kv, cl[kk][jt], cor);
        }
      cor = wave_sum(cor);
    }
    STAMP(3);
    f_res = (info == 0) ? 0.5 * ((double)n * LOG_2PI + logdet + ssq) + 0.5 * cor : nanv;
    info_res = info;
    STAMP_FLUSH();
    return;
  }
  abuf[lane] = alpha;
  wave_sync();  // abuf visible
  const BlockSums t = eval_products<NMAX>(lane, n, i, p, with_c, cpack, cs, Ks, tbuf, abuf);
  STAMP(7);
  eval_finish(n, sigma, logdet, ssq, info, t.s1, t.s2, t.s3, t.cor, f_res, g_res);
  info_res = info;
  STAMP(8);
  STAMP_FLUSH();
}

// Two wavefronts per SIMD where the instantiation's registers allow it (up to 52 rows).
template <int NMAX>
constexpr int indiv_eval_waves() { return NMAX <= 52 ? 2 : 1; }

template <int NMAX>
__global__ __launch_bounds__(64, MAGMA_INDIV_WAVES * indiv_eval_waves<NMAX>()) void indiv_eval_kernel(
    int nsel, const int* __restrict__ sel, const int64_t* __restrict__ off,
    const double* __restrict__ t, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, const double* __restrict__ mean,
    const double* __restrict__ cpack, int mode, int want_grad,
    double* __restrict__ f_out, double* __restrict__ g_out, int* __restrict__ info_out) {
  __shared__ double tbuf[64];
  __shared__ double zbuf[64];
  __shared__ double abuf[64];
  // Ks: KL<NMAX>::alloc doubles + slack for the fragment reads past a short last tile row. During
  // the sweep its first 64 NT doubles are the transposition buffer Pn[64][4].
  extern __shared__ __attribute__((aligned(16))) double Ks[];
  const int lane = threadIdx.x;
  const int io = blockIdx.x;  // slot of theta / f / g / info (compact when sel is given)
  const int i = sel ? sel[io] : io;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)io * tstride;
  const bool rv = lane < n;
  tbuf[lane] = rv ? t[o + lane] : 0.0;
  zbuf[lane] = rv ? (y[o + lane] - mean[idx[o + lane]]) : 0.0;
  __syncthreads();
  double f, g[3] = {0.0, 0.0, 0.0};
  int info;
  indiv_eval_body<NMAX>(lane, n, i, th[0], th[1], th[2], tbuf, zbuf, abuf, Ks, cpack, mode,
                        want_grad, f, g, info);
  if (lane == 0) {
    f_out[io] = f;
    info_out[io] = info;
    if (want_grad) {
      g_out[3 * (int64_t)io + 0] = g[0];
      g_out[3 * (int64_t)io + 1] = g[1];
      g_out[3 * (int64_t)io + 2] = g[2];
    }
  }
}

// ------------------------------------------------------------------ K1+K3: inverse to HBM
// Row `lane` of K^-1 from the lower-tile LDS image (KL): element (lane, c) for compile-time c sits
// in this lane's own tile row when c's tile is left of / on the diagonal tile, else in tile row
// c >> 4 by symmetry (see eval_alpha).
template <int NMAX>
struct KinvRow {
  int rt, rl, rsz, base_r;
  __device__ __forceinline__ explicit KinvRow(int lane) {
    using L = KL<NMAX>;
    const int lr = lane < NMAX ? lane : 0;
    rt = lr >> 4;
    rl = lr & 15;
    rsz = ((rt == L::NT - 1) ? L::rows(L::NT - 1) : 16) * L::TS;
    base_r = 16 * L::TS * (rt * (rt + 1) / 2) + rl * L::TS;
  }
  __device__ __forceinline__ double at(const double* Ks, int c) const {
    using L = KL<NMAX>;
    const int ct = c >> 4, cl = c & 15;
    const int a_row = base_r + ct * rsz + cl;
    const int a_col = L::tb(ct, 0) + rt * (L::rows(ct) * L::TS) + cl * L::TS + rl;
    return Ks[rt >= ct ? a_row : a_col];
  }
};

// With y / v given (the deterministic E-step): v_i = K_i^-1 y_i as well; a flagged (not positive
// definite) individual gets zeros in both here and its solve() / ginv() answer from
// small_slowpath_kernel, launched right behind this kernel. Same factorisation as the M-step's
// evaluations (eval_factor: lower tiles, two wavefronts per SIMD up to 52 rows).
template <int NMAX>
__global__ __launch_bounds__(64, indiv_eval_waves<NMAX>()) void se_inv_batched_kernel(
    int M, const int64_t* __restrict__ off, const int64_t* __restrict__ off2,
    const double* __restrict__ t, const double* __restrict__ theta, int tstride,
    double* __restrict__ out, double* __restrict__ logdet_out, int* __restrict__ info_out,
    const double* __restrict__ y, double* __restrict__ v) {
  __shared__ double tbuf[64];
  __shared__ double ybuf[64];
  extern __shared__ __attribute__((aligned(16))) double Ks[];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)i * tstride;
  const SEParams p = se_params(th[0], th[1], th[2]);
  tbuf[lane] = (lane < n) ? t[o + lane] : 0.0;
  ybuf[lane] = (v != nullptr && lane < n) ? y[o + lane] : 0.0;
  __syncthreads();
  double logdet;
  int info;
  eval_factor<NMAX>(lane, n, i, p, tbuf, Ks, nullptr, 0, logdet, info);
  double* Kinv = out + off2[i];
  const bool zero = (v != nullptr && info != 0);
  const KinvRow<NMAX> row(lane);
  if (lane < n) {
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c < n) {
        const double k = zero ? 0.0 : row.at(Ks, c);
        Kinv[(int64_t)c * n + lane] = k;
        s = fma(k, ybuf[c], s);
      }
    if (v != nullptr) v[o + lane] = s;
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
__global__ __launch_bounds__(64, indiv_eval_waves<NMAX>()) void estep_scatter_kernel(
    int M, const int64_t* __restrict__ off, const double* __restrict__ t,
    const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ theta, int tstride, double* __restrict__ P, int64_t ldp,
    double* __restrict__ w, int* __restrict__ info_out) {
  __shared__ double tbuf[64];
  __shared__ double ybuf[64];
  __shared__ int ibuf[64];
  extern __shared__ __attribute__((aligned(16))) double Ks[];
  const int i = blockIdx.x, lane = threadIdx.x;
  const int64_t o = off[i];
  const int n = (int)(off[i + 1] - o);
  const double* th = theta + (int64_t)i * tstride;
  const SEParams p = se_params(th[0], th[1], th[2]);
  const bool rv = lane < n;
  const int ir = rv ? idx[o + lane] : 0;
  tbuf[lane] = rv ? t[o + lane] : 0.0;
  ybuf[lane] = rv ? y[o + lane] : 0.0;
  ibuf[lane] = ir;
  __syncthreads();
  double logdet;
  int info;
  eval_factor<NMAX>(lane, n, i, p, tbuf, Ks, nullptr, 0, logdet, info);
  if (lane == 0) info_out[i] = info;
  if (info != 0) return;  // flagged: small_slowpath_kernel adds its solve() / ginv() answer
  const KinvRow<NMAX> row(lane);
  if (rv) {
    double wi = 0.0;
#pragma unroll
    for (int c = 0; c < NMAX; ++c) wi = fma(row.at(Ks, c), ybuf[c], wi);
    unsafeAtomicAdd(&w[ir], wi);
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      if (c <= lane) unsafeAtomicAdd(&P[(int64_t)ibuf[c] * ldp + ir], row.at(Ks, c));
    }
  }
}

// ------------------------------------------------------------------ device-resident M-step
// m_step's individual part (CF:419-432: one optim() per individual) without leaving the GPU:
// the wavefront that evaluates an individual also runs its L-BFGS-B (lbfgsb_core.h, the very
// code the host driver steps through, so both give the same iterates), evaluation after
// evaluation, until that problem stops. No host round trip per evaluation and no lock step: a
// wavefront whose problem has converged retires and the next individual takes its place, so
// the time is total work / machine width (or the longest single chain), not rounds x latency.
// The optimiser state (~60 doubles) lives in LDS; every lane steps it redundantly with the
// same wave-uniform f, g.
// Out of line on purpose: inlined, the optimiser's scalar code shares the register allocation of
// the evaluation body and pushes it into scratch spills (1 GB of spill traffic per launch at C3).
__device__ __attribute__((noinline)) void lbfgsb_tell_device(lbcore::State<3, 5>* st, double f,
                                                             double g0, double g1, double g2,
                                                             double factr, double pgtol,
                                                             int maxit) {
  const double g[3] = {g0, g1, g2};
  // (three variables here; five correction pairs is optim()'s default lmm)
  if (st->m == 5) lbcore::tell_fixed<3, 5>(*st, f, g, factr, pgtol, maxit);
  else lbcore::tell_fixed<3, 0>(*st, f, g, factr, pgtol, maxit);
}

// Workgroups of RW wavefronts, each wavefront an independent worker with its own slice of the
// workgroup's LDS. RW = 8 (two wavefronts per SIMD, 256 registers each) where eight slices fit the
// compute unit's 160 KB, else 4 (one per SIMD, all 512 registers): either way one workgroup fills
// one compute unit, and a launch of G workgroups occupies G compute units and leaves the others
// alone -- which is what the theta_0 chain's dense kernels running next to this one need (the
// workgroups of the dense inverse want a whole free compute unit). Individuals are handed out
// through a ticket counter in the order given (longest expected chain first): a wavefront that
// finishes a problem takes the next ticket.
//
// The tail. Which individuals need the long chains (100-200 evaluations against 5-8 on average)
// is not known ahead, so the launch used to last as long as the longest chain on one wavefront
// while the wavefronts next to it had nothing left to do. Now a wavefront that finds the tickets
// gone becomes a HELPER of its workgroup. The owner of a running problem opens its board: one claim
// word per tile row of K and per column block of stages 4 / 5. Helpers and the owner itself claim
// them (atomic exchange; block 0, the largest, first), each block's sums go to the owner's LDS,
// and the owner adds them in block order -- the order a single wavefront uses, so the result does
// not depend on who computed what. Helpers keep the packed C of the individual they last worked
// for in registers. All control flow is wave-uniform and in straight `if`s over compile-time block
// numbers (a run-time block number dispatched by a switch inside the claim loop compiled into code
// that never left the loop). Every wait is bounded and raises the workgroup's abort word, which
// turns the evaluation into a NaN (status 5 for that individual).
constexpr int RES_WAVES_MAX = 8;
constexpr int RC_NIDLE = 0, RC_ABORT = 1, RC_OWNER = 8, RC_STRIDE = 16;
constexpr int RO_CLAIM = 0 /* .. 3 */, RO_DONE = 4, RO_AREADY = 5, RO_EPOCH = 6, RO_I = 7, RO_N = 8,
              RO_LEFT = 9, RO_BCLAIM = 10 /* .. 13 */, RO_BDONE = 14, RO_BLEFT = 15;
constexpr int RC_WORDS = RC_OWNER + RC_STRIDE * RES_WAVES_MAX;
constexpr unsigned long long RES_HELPER_LIMIT = 300000000ull;  // 3 s of the 100 MHz clock

// One wavefront's LDS slice: Ks, timestamps, residuals, alpha, optimiser state, the block sums
// helpers leave ([block][s1, s2, s3, cor]).
template <int NMAX>
constexpr int resident_wave_doubles() {
  return KL<NMAX>::alloc + 3 * 64 + (int)((sizeof(lbcore::State<3, 5>) + 127) / 128) * 16 + 16;
}
template <int NMAX>
constexpr int resident_waves() {
  return (size_t)8 * resident_wave_doubles<NMAX>() * sizeof(double) + 4096 <= 160 * 1024 ? 8 : 4;
}

// Control words in LDS: one lane writes, every lane gets the same (scalar) value back.
__device__ __forceinline__ int ctl_load(const int* p) {
  return __builtin_amdgcn_readfirstlane(
      __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void ctl_store(int* p, int v, int lane) {
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ctl_add(int* p, int v, int lane) {
  if (lane == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool ctl_claim(int* p, int lane) {  // true: the word was 0 and is ours
  int v = 1;
  if (lane == 0) v = __hip_atomic_exchange(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __builtin_amdgcn_readfirstlane(v) == 0;
}

// Diagnostics of a workgroup that gave up on a wait (first writer wins): [0] code (1 owner waiting
// for block sums, 2 helper waiting for alpha, 3 helper loop beyond the time limit), [1] workgroup,
// [2] wavefront, [3] idle count, then done / alpha-ready / epoch / blocks left of the first four
// boards (code 4: owner waiting for the tile rows of K).
static __device__ unsigned long long g_res_diag[24];
__device__ __noinline__ void resident_diag(int code, int wave, const int* s_ctl) {
  if ((threadIdx.x & 63) != 0) return;
  if (atomicCAS(&g_res_diag[0], 0ull, (unsigned long long)code) != 0ull) return;
  g_res_diag[1] = blockIdx.x;
  g_res_diag[2] = wave;
  g_res_diag[3] = __hip_atomic_load(s_ctl + RC_NIDLE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int words[4] = {RO_DONE, RO_AREADY, RO_EPOCH, RO_LEFT};
  for (int o = 0; o < 4; ++o)
    for (int q = 0; q < 4; ++q)
      g_res_diag[4 + 4 * o + q] = (unsigned long long)(unsigned)__hip_atomic_load(
          s_ctl + RC_OWNER + RC_STRIDE * o + words[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The rest of an evaluation whose column blocks are on the board (see indiv_mstep_kernel). Out
// of line, with its LDS pointers typed as such: inlined next to the code of an evaluation that
// runs alone, it cost that one 2-3 % (spills and waits at the joins of the two).
#ifdef MAGMA_RES_STAMPS
__device__ unsigned long long g_res_stamps[16];
#define RSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_res_stamps[k], t_ - rs_prev); rs_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define RSTAMP_INIT() unsigned long long rs_prev = __builtin_readcyclecounter()
#else
#define RSTAMP(k)
#define RSTAMP_INIT()
#endif
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
struct HelpedSums {
  double s1, s2, s3, cor, ssq, logdet;
  int info, ok;
};
// One claimed column block, by the owner or by a helper: T[:, jt] from the wavefront's copy of the
// packed C, stage 5, the block's sums into the owner's slice.
template <int NMAX, int JT, class WaitAlpha>
__device__ __forceinline__ void resident_do_block(int lane, int n, const SEParams& p, bool with_c,
                                                  const double (&cf)[NMAX / 4][(NMAX + 15) / 16],
                                                  const double* Ks, const double* tbuf,
                                                  const double* abuf, double* part,
                                                  WaitAlpha&& wait_alpha) {
  constexpr int NT = (NMAX + 15) / 16;
  v4d T[NT][NT];
  if (with_c) eval_tblock<NMAX, JT>(lane, cf, Ks, T);
  double q1, q2, q3, qc;
  eval_zblock<NMAX, JT>(lane, n, p, with_c, T, Ks, tbuf, abuf, wait_alpha, q1, q2, q3, qc);
  const BlockSums b = block_reduce(q1, q2, q3, qc);
  if (lane == 0) {
    part[4 * JT + 0] = b.s1;
    part[4 * JT + 1] = b.s2;
    part[4 * JT + 2] = b.s3;
    part[4 * JT + 3] = b.cor;
  }
}

template <int NMAX>
__device__ __attribute__((noinline)) HelpedSums resident_helped_eval(
    lds_double* Ks_, lds_int* s_ctl_, lds_double* s_par_, int wave, int lane, int n, int i, int epoch,
    SEParams p, double zr, const double* __restrict__ cpack) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  double* Ks = (double*)Ks_;
  int* s_ctl = (int*)s_ctl_;
  double* s_parw = (double*)s_par_ + 4 * wave;
  double* tbuf = Ks + KL<NMAX>::alloc;
  double* zbuf = tbuf + 64;
  double* abuf = zbuf + 64;
  double* part = Ks + resident_wave_doubles<NMAX>() - 16;
  int* ctl = s_ctl + RC_OWNER + RC_STRIDE * wave;
  int* abort_ = s_ctl + RC_ABORT;
  const bool with_c = (cpack != nullptr);
  HelpedSums r;
  RSTAMP_INIT();
  // ---- the board; K and K^-1 by this wavefront (round 3: with the lower-tile build -- 28 exp per
  // lane -- handing the tile rows of K to the helpers costs more in claims, fences and polls than
  // the rows themselves: 9.7k cycles on the board against 7.6k alone)
  if (lane == 0) {
    s_parw[0] = p.th1;
    s_parw[1] = p.h;
    s_parw[2] = p.sig2;
    s_parw[3] = p.eth1;
  }
  ctl_store(ctl + RO_I, i, lane);
  ctl_store(ctl + RO_N, n, lane);
  ctl_store(ctl + RO_EPOCH, epoch, lane);
  ctl_store(ctl + RO_DONE, 0, lane);
  bool ok = true;
  RSTAMP(5);
  eval_factor<NMAX, false>(lane, n, i, p, tbuf, Ks, nullptr, 0, r.logdet, r.info);
  RSTAMP(6);

  // ---- the column blocks of stages 4 / 5
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  // K^-1 first
  ctl_store(ctl + RO_LEFT, NT, lane);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) ctl_store(ctl + RO_CLAIM + jt, 0, lane);
  RSTAMP(0);

  const double alpha = eval_alpha<NMAX>(lane, Ks, zbuf);
  r.ssq = wave_sum(zr * alpha);
  abuf[lane] = alpha;
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  ctl_store(ctl + RO_AREADY, epoch, lane);
  RSTAMP(1);

  // the helpers have had the time of alpha to claim the blocks; what is left is this wavefront's
  // (the packed C is only loaded if there is something left)
  if (ctl_load(ctl + RO_LEFT) > 0) {
    bool have_c = false;
    double cf[NK][NT];
    for_blocks<NMAX>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      if (ctl_claim(ctl + RO_CLAIM + jt, lane)) {
        if (with_c && !have_c) {
          __builtin_amdgcn_sched_barrier(0);
          eval_load_c<NMAX>(cf, cpack, i, lane);
          __builtin_amdgcn_sched_barrier(0);
          have_c = true;
        }
        resident_do_block<NMAX, jt>(lane, n, p, with_c, cf, Ks, tbuf, abuf, part, [] {});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        ctl_add(ctl + RO_LEFT, -1, lane);
        ctl_add(ctl + RO_DONE, 1, lane);
      }
    });
  }
  RSTAMP(2);
  if (!lds_wait_ge(ctl + RO_DONE, NT, abort_)) {
    resident_diag(1, wave, s_ctl);
    ok = false;
  }
  RSTAMP(3);
  r.s1 = r.s2 = r.s3 = r.cor = 0.0;
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    r.s1 += part[4 * jt + 0];
    r.s2 += part[4 * jt + 1];
    r.s3 += part[4 * jt + 2];
    r.cor += part[4 * jt + 3];
  }
  r.ok = ok ? 1 : 0;
  wave_sync();
  RSTAMP(4);
#ifdef MAGMA_RES_STAMPS
  if (lane == 0) atomicAdd(&g_res_stamps[15], 1ull);
#endif
  return r;
}

template <int NMAX>
__global__ __launch_bounds__(64 * resident_waves<NMAX>(), resident_waves<NMAX>() / 4) void indiv_mstep_kernel(
    int M, const int* __restrict__ order, const int64_t* __restrict__ off,
    const double* __restrict__ t, const double* __restrict__ y, const int* __restrict__ idx,
    const double* __restrict__ mean, const double* cpack, int mode,
    const double* __restrict__ theta0, int lm,
    double factr, double pgtol, int maxit, double* __restrict__ theta_out,
    double* __restrict__ f_out, double* __restrict__ fx0_out, int* __restrict__ iters_out,
    int* __restrict__ nfev_out, int* __restrict__ status_out, int* __restrict__ ticket,
    int help_on, int slow_all, const double* __restrict__ gather_cov, int64_t gather_ld) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4;
  constexpr int RW = resident_waves<NMAX>();
  static_assert(NT <= 4, "four claim words per board");
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  __shared__ int s_ctl[RC_WORDS];
  __shared__ double s_par[RES_WAVES_MAX][4];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar: LDS bases in SGPRs
  double* Ks = dyn + (size_t)wave * resident_wave_doubles<NMAX>();
  double* tbuf = Ks + KL<NMAX>::alloc;
  double* zbuf = tbuf + 64;
  double* abuf = zbuf + 64;
  lbcore::State<3, 5>* st = reinterpret_cast<lbcore::State<3, 5>*>(abuf + 64);
  int* ctl = s_ctl + RC_OWNER + RC_STRIDE * wave;
  int* abort_ = s_ctl + RC_ABORT;
  (void)ctl;
  if (threadIdx.x < RC_WORDS) {  // boards closed (every block claimed), everything else zero
    const int q = ((int)threadIdx.x - RC_OWNER) % RC_STRIDE;
    const bool claim = (q >= RO_CLAIM && q < RO_CLAIM + 4) || (q >= RO_BCLAIM && q < RO_BCLAIM + 4);
    s_ctl[threadIdx.x] = ((int)threadIdx.x >= RC_OWNER && claim) ? 1 : 0;
  }
  __syncthreads();
  const bool with_c = (mode == 0 && cpack != nullptr);
  int i = 0, n = 0, epoch = 0;
  bool fresh = true;
  double fx0 = 0.0;
#pragma unroll 1
  for (;;) {
    if (fresh) {  // wave-uniform
      int slot = 0;
      if ((threadIdx.x & 63) == 0) slot = atomicAdd(ticket, 1);
      slot = __builtin_amdgcn_readfirstlane(slot);
      if (slot >= M) break;
      i = order ? order[slot] : slot;
      wave_sync();
      const int64_t o = off[i];
      n = (int)(off[i + 1] - o);
      const int lane0 = threadIdx.x & 63;
      const bool rv = lane0 < n;
      tbuf[lane0] = rv ? t[o + lane0] : 0.0;
      zbuf[lane0] = rv ? (y[o + lane0] - mean[idx[o + lane0]]) : 0.0;
      lbcore::init(*st, 3, lm, theta0 + 3 * (int64_t)i);
      wave_sync();
      if (gather_cov != nullptr && cpack != nullptr) {
        // This individual's block C_i = cov[idx_i, idx_i] into its slot of the packed array
        // (layout of indiv_gather_cov_kernel), by the wavefront that is about to use it: the
        // separate gather launch (0.2 ms between the E-step and this kernel) is gone, its reads
        // are spread under the other wavefronts' evaluations. cov is symmetric in memory.
        int* ib = reinterpret_cast<int*>(abuf);
        ib[lane0] = rv ? idx[o + lane0] : 0;
        wave_sync();
        constexpr int TOT = 16 * NT * NMAX;
        double* out = const_cast<double*>(cpack) + (int64_t)i * TOT;
#pragma unroll 13
        for (int e = lane0; e < TOT; e += 64) {
          const int it = e % NT, ln = (e / NT) & 63, kk = e / (64 * NT);
          const int r = 16 * it + (ln & 15), c = 4 * kk + (ln >> 4);
          const bool in = r < n && c < n;
          const double v = gather_cov[(int64_t)ib[in ? c : 0] * gather_ld + ib[in ? r : 0]];
          out[e] = in ? v : 0.0;
        }
        __threadfence();  // the block is read back through other instructions (and by helpers)
        wave_sync();
      }
      fresh = false;
    }
    // lane is laundered per evaluation: otherwise every lane-derived address and mask of the
    // body is hoisted out of the loop and held in registers across it
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const double th1 = st->xtrial[0], th2 = st->xtrial[1], sigma = st->xtrial[2];
    const SEParams p = se_params(th1, th2, sigma);
    const double zr = zbuf[lane];
    double logdet;
    int info;
    // Somebody idle in this workgroup? Then this evaluation goes on the board (mode 0 only: K
    // without C). Its code is a function of its own: sharing the code between the two cases costs
    // the one that runs alone 3 % -- spills and waits at the joins.
    const bool helped = help_on != 0 && mode == 0 && ctl_load(s_ctl + RC_NIDLE) > 0;
    BlockSums tot = {0.0, 0.0, 0.0, 0.0};
    double ssq;
#ifdef MAGMA_RES_STAMPS
    const unsigned long long rs_e0 = __builtin_readcyclecounter();
#endif
    if (!helped) {
      eval_factor<NMAX>(lane, n, i, p, tbuf, Ks, cpack, mode, logdet, info);
      __builtin_amdgcn_sched_barrier(0);
      CStream<NMAX> cs;
      if (with_c) eval_cstream_start<NMAX>(cs, lane, i, cpack);
      __builtin_amdgcn_sched_barrier(0);
      const double alpha = eval_alpha<NMAX>(lane, Ks, zbuf);
      ssq = wave_sum(zr * alpha);
      abuf[lane] = alpha;
      wave_sync();  // abuf visible
      tot = eval_products<NMAX>(lane, n, i, p, with_c, cpack, cs, Ks, tbuf, abuf);
    } else {
      ++epoch;
      const HelpedSums r = resident_helped_eval<NMAX>((lds_double*)Ks, (lds_int*)s_ctl,
                                                      (lds_double*)&s_par[0][0], wave, lane, n, i,
                                                      epoch, p, zr, cpack);
      tot.s1 = r.s1;
      tot.s2 = r.s2;
      tot.s3 = r.s3;
      tot.cor = r.cor;
      ssq = r.ssq;
      logdet = r.logdet;
      info = r.ok ? r.info : -999;
    }
    double f, g[3] = {0.0, 0.0, 0.0};
    eval_finish(n, sigma, logdet, ssq, info, tot.s1, tot.s2, tot.s3, tot.cor, f, g);

    fx0 = (st->nfev == 0) ? f : fx0;  // (a loop counter here makes hipcc peel the first pass)
    wave_sync();  // the body's LDS traffic is over before the state is stepped
#ifdef MAGMA_RES_STAMPS
    const unsigned long long rs_t0 = __builtin_readcyclecounter();
#endif
    lbfgsb_tell_device(st, f, g[0], g[1], g[2], factr, pgtol, maxit);
    // A covariance the sweep flagged: the reference evaluates it through solve() / ginv()
    // (CF:133); that evaluation lives in indiv_eval_slowpath_kernel. At the START point the
    // problem stops with status 6 and the host solves it through the lock-step route, whose
    // evaluations take that path (the optimiser cannot start from a non-finite value). At a
    // trial point of a line search the non-finite value makes the search bisect towards its
    // best point (lbfgsb_core.h) -- unless slow_all asks for status 6 there too (option
    // "resident_slow_path": the reference's value at every flagged point, at the price of
    // solving those problems twice; see DESIGN.md for the measured cost).
    if (info != 0 && info != -999 && (slow_all != 0 || st->nfev == 1)) st->status = 6;
#ifdef MAGMA_RES_STAMPS
    if (helped && lane == 0) atomicAdd(&g_res_stamps[12], __builtin_readcyclecounter() - rs_t0);  // tell
    if (helped && lane == 0) atomicAdd(&g_res_stamps[13], rs_t0 - rs_e0);  // start -> tell
#endif
    wave_sync();
#ifdef MAGMA_RES_STAMPS
    if (helped && lane == 0) {
      const unsigned long long rs_e2 = __builtin_readcyclecounter();
      atomicAdd(&g_res_stamps[9], rs_e2 - rs_e0);   // whole evaluation + optimiser step
    }
#endif
    if (st->status != 0) {
      if ((threadIdx.x & 63) == 0) {
        const double nanv = __longlong_as_double(0x7ff8000000000000LL);
        for (int q = 0; q < 3; ++q)
          theta_out[3 * (int64_t)i + q] = (st->status == 5) ? nanv : st->x[q];
        f_out[i] = st->f;
        fx0_out[i] = fx0;
        iters_out[i] = st->iter;
        nfev_out[i] = st->nfev;
        status_out[i] = st->status;
      }
      fresh = true;
    }
  }

  // ---- no tickets left: help the wavefronts of this workgroup that still run a problem
  if (help_on == 0) return;
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  ctl_add(s_ctl + RC_NIDLE, 1, lane);
  int tag = -1;  // individual whose packed C is in cf
  const unsigned long long t_idle = wall_clock64();  // 100 MHz
  double cf[NK][NT];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int it = 0; it < NT; ++it) cf[kk][it] = 0.0;
#pragma unroll 1
  for (;;) {
    if (ctl_load(abort_) != 0) break;
    if (ctl_load(s_ctl + RC_NIDLE) >= RW) break;
    bool did = false;
#pragma unroll 1
    for (int q = 1; q < RW; ++q) {
      const int o = (wave + q) & (RW - 1);
      int* co = s_ctl + RC_OWNER + RC_STRIDE * o;
      double* oKs = dyn + (size_t)o * resident_wave_doubles<NMAX>();
      const double* otbuf = oKs + KL<NMAX>::alloc;
      const double* oabuf = otbuf + 128;
      double* opart = oKs + resident_wave_doubles<NMAX>() - 16;
      if (ctl_load(co + RO_LEFT) <= 0) continue;
      for_blocks<NMAX>([&](auto JT) {
        constexpr int jt = decltype(JT)::value;
        if (ctl_claim(co + RO_CLAIM + jt, lane)) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
          const int oi = ctl_load(co + RO_I);
          const int on = ctl_load(co + RO_N);
          const int oe = ctl_load(co + RO_EPOCH);
          SEParams p;
          p.th1 = s_par[o][0];
          p.h = s_par[o][1];
          p.sig2 = s_par[o][2];
          p.eth1 = s_par[o][3];
          if (with_c && tag != oi) {
            __builtin_amdgcn_sched_barrier(0);
            eval_load_c<NMAX>(cf, cpack, oi, lane);
            __builtin_amdgcn_sched_barrier(0);
            tag = oi;
          }
          resident_do_block<NMAX, jt>(lane, on, p, with_c, cf, oKs, otbuf, oabuf, opart, [&] {
            if (!lds_wait_ge(co + RO_AREADY, oe, abort_)) resident_diag(2, wave, s_ctl);
          });
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
          ctl_add(co + RO_LEFT, -1, lane);
          ctl_add(co + RO_DONE, 1, lane);
          did = true;
        }
      });
    }
    if (!did) {
      // (nothing in this launch lasts seconds: a helper that has waited that long gives up)
      if (wall_clock64() - t_idle > RES_HELPER_LIMIT) {
        resident_diag(3, wave, s_ctl);
        ctl_store(abort_, 1, lane);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

// ------------------------------------------------------------------ prediction (MG:172-180)
// One wavefront inverts cov_tn_tn = K(tn) + C[in,in]; each workgroup then handles a chunk
// of target timestamps: Mean = m_t + k' inv (y - m_obs), Var = (e^th1 + sigma^2 + C[t,t])
// - k' inv k, with k = kern(tn - t) + C[in, t] (plain kernel, no noise: MG:174).
#define PRED_GP_LDS                                            \
  __shared__ double tbuf[64];                                  \
  __shared__ double zbuf[64];                                  \
  __shared__ double kbuf[64];                                  \
  __shared__ int ibuf[64];                                     \
  extern __shared__ __attribute__((aligned(16))) double Ks[]

// cov_tn_tn = K(tn) + C[in, in] (MG:172-173; C = 0 without cov) in accumulator layout, scaled to
// a diagonal <= 1, inverted with the blocked MFMA sweep (mfma_sweep.h), left in LDS as
// Ks[r * NMAX + c] (symmetric; NMAX * NMAX + 16 doubles). Ends with a wave_sync().
template <int NMAX>
__device__ __forceinline__ void pred_inverse_to_lds(int lane, int n, const SEParams& p,
                                                    const double* tbuf, const int* ibuf,
                                                    const double* __restrict__ cov, int64_t ldc,
                                                    double* Ks, int& info) {
  constexpr int NT = (NMAX + 15) / 16, NK = NMAX / 4, LD = NMAX;
  static_assert(NMAX % 4 == 0, "NMAX must be a multiple of 4");
  const int l = lane & 15, g = lane >> 4;
  v4d A[NT][NT];
  double tcol[NT];
  int icol[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    tcol[jt] = tbuf[16 * jt + l];
    icol[jt] = ibuf[16 * jt + l];
  }
  double dm = 0.0;
#pragma unroll
  for (int it = 0; it < NT; ++it) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * it + 4 * v >= NMAX) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) A[it][jt][v] = 0.0;
        continue;
      }
      const int r = 16 * it + 4 * v + g;
      const double trow = tbuf[r];
      const int irow = ibuf[r];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int c = 16 * jt + l;
        const bool in = r < n && c < n;
        const double d = trow - tcol[jt];
        double kv = exp(p.th1 - p.h * d * d);
        kv = (r == c) ? (p.eth1 + p.sig2) : kv;
        if (cov != nullptr && in) kv += cov[(int64_t)icol[jt] * ldc + irow];
        A[it][jt][v] = in ? kv : 0.0;
        dm = fmax(dm, (in && r == c) ? kv : 0.0);
        __builtin_amdgcn_sched_barrier(0);  // keep the unrolled exp() bodies sequential
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dm = fmax(dm, __shfl_xor(dm, o));
  // (a non-positive diagonal cannot be scaled; the sweep then flags the matrix by itself)
  const double sc = pow2_scale(dm);
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * it + 4 * v >= NMAX) continue;
      const int r = 16 * it + 4 * v + g;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int c = 16 * jt + l;
        A[it][jt][v] = (r < n && c < n) ? A[it][jt][v] * sc : ((r == c) ? 1.0 : 0.0);
      }
    }
  double pm = 1.0;
  int pe = 0;
  info = 0;
  wave_sync();
  mfma_sweep<NT, NK>(A, Ks, l, g, pm, pe, info);
  {  // (see eval_factor: a stand-in for the rcond test of R's solve())
    double dinv = 0.0;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (16 * it + 4 * v < NMAX) dinv = fmax(dinv, (g + 4 * v == l) ? -A[it][it][v] : 0.0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dinv = fmax(dinv, __shfl_xor(dinv, o));
    info = (info == 0 && !(dinv < SWEEP_INVDIAG_MAX)) ? n + 1 : info;
  }
#pragma unroll
  for (int it = 0; it < NT; ++it)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (16 * it + 4 * v >= NMAX) continue;
      const int r = 16 * it + 4 * v + g;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int c = 16 * jt + l;
        if (16 * jt + 15 < NMAX || c < NMAX) Ks[r * LD + c] = -sc * A[it][jt][v];
      }
    }
  wave_sync();
}

// mean_grid != nullptr: the mean process is given on cov's grid and gathered through idx_n /
// idx_t (the device-resident route of the batched entry point); else mean_obs / mean_pred hold it.
template <int NMAX, bool GRID>
__device__ __forceinline__ void pred_gp_body(
    int n, const double* __restrict__ tn, const double* __restrict__ yn,
    const double* __restrict__ mean_obs, const int* __restrict__ idx_n, int T,
    const double* __restrict__ ts, const double* __restrict__ mean_pred,
    const int* __restrict__ idx_t, const double* __restrict__ mean_grid,
    const double* __restrict__ cov, int64_t ldc,
    const double* __restrict__ theta, double* __restrict__ out_mean,
    double* __restrict__ out_var, int* __restrict__ info_out, int chunk_id, int chunk,
    double* tbuf, double* zbuf, double* kbuf, int* ibuf, double* Ks) {
  const int lane = threadIdx.x;
  const double sigma = theta[2];
  const SEParams p = se_params(theta[0], theta[1], sigma);
  const bool rv = lane < n;
  const double tr = rv ? tn[lane] : 0.0;
  const int ir = (rv && idx_n) ? idx_n[lane] : 0;
  double mo = 0.0;
  if constexpr (GRID) {
    if (rv) mo = mean_grid[ir];
  } else {
    if (rv) mo = mean_obs[lane];
  }
  const double zr = rv ? (yn[lane] - mo) : 0.0;
  tbuf[lane] = tr;
  zbuf[lane] = zr;
  ibuf[lane] = ir;
  __syncthreads();
  int info;
  pred_inverse_to_lds<NMAX>(lane, n, p, tbuf, ibuf, cov, ldc, Ks, info);
  if (lane == 0 && chunk_id == 0) info_out[0] = info;
  // lane r owns row r = column r of the inverse: Ks[c * NMAX + r] is conflict free
  const int lr = lane < NMAX ? lane : 0;
  double alpha = 0.0;
#pragma unroll
  for (int c = 0; c < NMAX; ++c) alpha = fma(Ks[c * NMAX + lr], zbuf[c], alpha);
  alpha = rv ? alpha : 0.0;
  const int j0 = chunk_id * chunk;
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
    for (int c = 0; c < NMAX; ++c) u = fma(Ks[c * NMAX + lr], kbuf[c], u);
    const double q = wave_sum(kv * u);
    const double m = wave_sum(kv * alpha);
    if (lane == 0) {
      const double ctt = (cov != nullptr) ? cov[(int64_t)it * ldc + it] : 0.0;
      double mp;
      if constexpr (GRID) mp = mean_grid[it]; else mp = mean_pred[j];
      out_mean[j] = mp + m;
      out_var[j] = (p.eth1 + p.sig2 + ctt) - q;
    }
  }
}

template <int NMAX>
__global__ __launch_bounds__(64) void pred_gp_kernel(
    int n, const double* __restrict__ tn, const double* __restrict__ yn,
    const double* __restrict__ mean_obs, const int* __restrict__ idx_n, int T,
    const double* __restrict__ ts, const double* __restrict__ mean_pred,
    const int* __restrict__ idx_t, const double* __restrict__ cov, int64_t ldc,
    const double* __restrict__ theta, double* __restrict__ out_mean,
    double* __restrict__ out_var, int* __restrict__ info_out, int chunk) {
  PRED_GP_LDS;
  pred_gp_body<NMAX, false>(n, tn, yn, mean_obs, idx_n, T, ts, mean_pred, idx_t, nullptr, cov, ldc, theta,
                     out_mean, out_var, info_out, blockIdx.x, chunk, tbuf, zbuf, kbuf, ibuf, Ks);
}

// pred_gp for B new individuals against ONE device-resident hyper-posterior (the
// step after training, `loop_pred` / `full_algo` per new individual): workgroup (chunk, b)
// handles `chunk` targets of individual b. mean / cov stay in HBM where posterior_mu left them;
// only the ragged observations, targets and their grid positions come in.
template <int NMAX>
__global__ __launch_bounds__(64) void pred_gp_batched_kernel(
    const int64_t* __restrict__ off_n, const double* __restrict__ tn,
    const double* __restrict__ yn, const int* __restrict__ idx_n,
    const int64_t* __restrict__ off_t, const double* __restrict__ ts,
    const int* __restrict__ idx_t, const double* __restrict__ mean_grid,
    const double* __restrict__ cov, int64_t ldc, const double* __restrict__ theta, int tstride,
    double* __restrict__ out_mean, double* __restrict__ out_var, int* __restrict__ info_out,
    int chunk) {
  PRED_GP_LDS;
  const int b = blockIdx.y;
  const int64_t on = off_n[b], ot = off_t[b];
  const int n = (int)(off_n[b + 1] - on), T = (int)(off_t[b + 1] - ot);
  if ((int)blockIdx.x * chunk >= T && blockIdx.x != 0) return;  // (chunk 0 always reports info)
  pred_gp_body<NMAX, true>(n, tn + on, yn + on, nullptr, idx_n + on, T, ts + ot, nullptr, idx_t + ot,
                     mean_grid, cov, ldc, theta + (int64_t)b * tstride, out_mean + ot, out_var + ot,
                     info_out + b, blockIdx.x, chunk, tbuf, zbuf, kbuf, ibuf, Ks);
}

// ------------------------------------------------------------------ launchers
// (the instantiation list is mirrored by small_template_nmax() in common.h)
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

// hipFuncSetAttribute is per device and the library may be driven by one thread per GPU: the
// guard is one atomic flag per device and kernel instantiation (setting it twice is harmless).
struct AttrOnce {
  std::atomic<bool> done[64];
  AttrOnce() { for (auto& d : done) d.store(false); }
  template <class F>
  int ensure(F&& set) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (done[dev].load(std::memory_order_acquire)) return 0;
    if (int rc = set()) return rc;
    done[dev].store(true, std::memory_order_release);
    return 0;
  }
};

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
                          double* logdet, int* info, const double* y, double* v, hipStream_t st) {
  if (M <= 0) return 0;
#define CALL(NM)                                                                              \
  hipLaunchKernelGGL(se_inv_batched_kernel<NM>, dim3(M), dim3(64),                              \
                     (KL<NM>::alloc + 16 * KL<NM>::TS) * sizeof(double), st, M, off, off2, t, theta, \
                     tstride, out, logdet, info, y, v)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return check_hip(hipGetLastError(), "se_inv_batched");
}

int launch_estep_scatter(int M, int nmax, const int64_t* off, const double* t, const double* y,
                         const int* idx, const double* theta, int tstride, double* P,
                         int64_t ldp, double* w, int* info, hipStream_t st) {
  if (M <= 0) return 0;
#define CALL(NM)                                                                             \
  hipLaunchKernelGGL(estep_scatter_kernel<NM>, dim3(M), dim3(64),                             \
                     (KL<NM>::alloc + 16 * KL<NM>::TS) * sizeof(double), st, M, off, t, y, idx,    \
                     theta, tstride, P, ldp, w, info)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return check_hip(hipGetLastError(), "estep_scatter");
}

// Dynamic LDS above 64 KB needs the attribute once per device and kernel (which: 0..3).
static int slowpath_lds_attr(const void* fn, size_t dyn, int which) {
  static AttrOnce once[4];
  return once[which].ensure([&] {
    return check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn),
                     "slow path LDS size");
  });
}

int launch_small_slowpath(int M, const int64_t* off, const int64_t* off2, const double* t,
                          const double* y, const int* idx, const double* theta, int tstride,
                          const int* info, double* Kinv, double* v, double* logdet, double* P,
                          int64_t ldp, double* w, int* route, hipStream_t st) {
  if (M <= 0) return 0;
  const size_t dyn = (size_t)SP_LDS_DOUBLES * sizeof(double);
  if (int rc = slowpath_lds_attr((const void*)small_slowpath_kernel, dyn, 0)) return rc;
  const int grid = M < 512 ? M : 512;  // unflagged individuals cost one load each
  hipLaunchKernelGGL(small_slowpath_kernel, dim3(grid), dim3(64), dyn, st, M, off, off2, t, y, idx,
                     theta, tstride, info, Kinv, v, logdet, P, ldp, w, route);
  return check_hip(hipGetLastError(), "small_slowpath");
}

int launch_csr_gather_sum(int64_t ntgt, const int64_t* ptr, const int* src, const int64_t* tgt,
                          const double* val, double* out, int wave_per_target, hipStream_t st) {
  if (ntgt <= 0) return 0;
  if (wave_per_target) {
    const int64_t g = (ntgt + 3) / 4;
    hipLaunchKernelGGL(csr_gather_sum_wave_kernel, dim3((unsigned)g), dim3(256), 0, st, ntgt, ptr,
                       src, tgt, val, out);
    return check_hip(hipGetLastError(), "csr_gather_sum_wave");
  }
  const int64_t grid = (ntgt + 255) / 256;
  // 16 consecutive blocks per XCD: FETCH_SIZE 466 -> 208 MB at C3, 298 -> 268 us alone (larger
  // chunks fetch a little less and are no faster; one contiguous eighth per XCD is slower, 361 us)
  const int lg = 4;
  hipLaunchKernelGGL(csr_gather_sum_kernel, dim3((unsigned)grid), dim3(256), 0, st, ntgt, ptr, src,
                     tgt, val, out, lg);
  return check_hip(hipGetLastError(), "csr_gather_sum");
}

int launch_update_inv(int M, const int64_t* off, const int64_t* off2, const int* idx,
                      const double* Kinv, const double* y, double* P, int64_t ldp, double* w,
                      hipStream_t st) {
  if (M <= 0) return 0;
  const int grid = M < 65536 ? M : 65536;
  hipLaunchKernelGGL(update_inv_kernel, dim3(grid), dim3(256), 0, st, M, off, off2, idx, Kinv, y, P,
                     ldp, w);
  return check_hip(hipGetLastError(), "update_inv");
}

template <int NM>
static int launch_indiv_eval_t(int nsel, const int* sel, const int64_t* off, const double* t,
                               const double* y, const int* idx, const double* theta, int tstride,
                               const double* mean, const double* cpack, int mode, int want_grad,
                               double* f, double* g, int* info, hipStream_t st) {
  // Ks + what the fragment reads of a short last tile row run past it (KL)
  const size_t dyn = ((size_t)KL<NM>::alloc + 16 * KL<NM>::TS) * sizeof(double);
  static AttrOnce once;
  if (int rc = once.ensure([&] {
        return check_hip(hipFuncSetAttribute((const void*)indiv_eval_kernel<NM>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn),
                         "indiv_eval LDS size");
      }))
    return rc;
  hipLaunchKernelGGL(indiv_eval_kernel<NM>, dim3(nsel), dim3(64), dyn, st, nsel, sel, off, t, y,
                     idx, theta, tstride, mean, cpack, mode, want_grad, f, g, info);
  return check_hip(hipGetLastError(), "indiv_eval");
}

int launch_indiv_eval(int nsel, const int* sel, int nmax, const int64_t* off, const double* t,
                      const double* y, const int* idx, const double* theta, int tstride,
                      const double* mean, const double* cpack, int mode, int want_grad,
                      double* f, double* g, int* info, hipStream_t st) {
  if (nsel <= 0) return 0;
  int rc = 0;
#define CALL(NM)                                                                          \
  rc = launch_indiv_eval_t<NM>(nsel, sel, off, t, y, idx, theta, tstride, mean, cpack,   \
                               mode, want_grad, f, g, info, st)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  if (rc) return rc;
  // the slots the sweep flagged: solve() / ginv(), as the reference's tryCatch does
  const size_t dyn = (size_t)SP_LDS_DOUBLES * sizeof(double);
  if (int r2 = slowpath_lds_attr((const void*)indiv_eval_slowpath_kernel, dyn, 1)) return r2;
  hipLaunchKernelGGL(indiv_eval_slowpath_kernel, dim3((nsel + 63) / 64), dim3(64), dyn, st, nsel, sel,
                     off, t, y, idx, theta, tstride, mean, cpack, cpack_ldp(nmax) / 16,
                     small_template_nmax(nmax), mode, want_grad, f, g, info);
  return check_hip(hipGetLastError(), "indiv_eval slow path");
}

// Cap on the workgroups (= compute units) of the device-resident M-step; 0 = every compute unit.
// Per host thread: the driver sets it around its own launch (one thread per GPU; several ranks as
// threads of one process must not clear each other's split), the option sets the default of every
// thread that has not set its own.
static std::atomic<int> g_resident_max_wgs{0};
static thread_local int t_resident_max_wgs = -1;
void set_resident_max_workgroups(int v) { g_resident_max_wgs.store(v < 0 ? 0 : v); }
void set_resident_max_workgroups_thread(int v) { t_resident_max_wgs = v; }
// The next resident launch of this host thread gathers every individual's packed cov block from
// `cov` itself (instead of expecting magma_indiv_gather_cov to have run): the driver's route.
static thread_local const double* t_resident_gather_cov = nullptr;
static thread_local int64_t t_resident_gather_ld = 0;
void set_resident_gather_thread(const double* cov, int64_t ld) {
  t_resident_gather_cov = cov;
  t_resident_gather_ld = ld;
}
static std::atomic<bool> g_resident_gather{true};
bool resident_gather_enabled() { return g_resident_gather.load(); }
void set_resident_gather(bool on) { g_resident_gather.store(on); }
// Idle wavefronts help the running ones of their workgroup (default on; same results either way).
static std::atomic<int> g_resident_help{1};
void set_resident_help(int v) { g_resident_help.store(v & 3); }
// 1: a flagged trial point inside a line search also stops the problem with status 6 (default 0:
// only a flagged start point does; a flagged trial point bisects)
static std::atomic<int> g_resident_slow{0};
void set_resident_slow_path(int v) { g_resident_slow.store(v ? 1 : 0); }  // bit 1: every compute unit even for few individuals (diagnostic)
int resident_diag_read(unsigned long long* h24) {
#ifdef MAGMA_RES_STAMPS
  return check_hip(hipMemcpyFromSymbol(h24, HIP_SYMBOL(g_res_stamps), 16 * sizeof(unsigned long long)),
                   "resident stamps");
#else
  return check_hip(hipMemcpyFromSymbol(h24, HIP_SYMBOL(g_res_diag), 24 * sizeof(unsigned long long)),
                   "resident diagnostics");
#endif
}

// Ticket counters of the resident launches: a small ring in device memory, one word per launch
// (zeroed on the launch's stream), so that launches running at the same time on one GPU -- several
// engines, several streams -- never share one. A word is reused after RES_TICKETS launches.
constexpr int RES_TICKETS = 1024;
static __device__ int g_res_tickets[RES_TICKETS];
static std::atomic<unsigned> g_res_ticket_next{0};

static int resident_ticket(int** out, hipStream_t st) {
  static std::mutex mu;
  static int* base[64] = {nullptr};
  static int ncu[64] = {0};
  int dev = 0;
  if (int rc = check_hip(hipGetDevice(&dev), "indiv_mstep")) return rc;
  if (dev < 0 || dev >= 64) return fail(2, "indiv_mstep: device index out of range");
  {
    std::lock_guard<std::mutex> lk(mu);
    if (base[dev] == nullptr) {
      void* p = nullptr;
      if (int rc = check_hip(hipGetSymbolAddress(&p, HIP_SYMBOL(g_res_tickets)), "indiv_mstep")) return rc;
      base[dev] = static_cast<int*>(p);
    }
  }
  (void)ncu;
  *out = base[dev] + (g_res_ticket_next.fetch_add(1) % RES_TICKETS);
  return check_hip(hipMemsetAsync(*out, 0, sizeof(int), st), "indiv_mstep ticket");
}

static int device_compute_units() {
  static std::atomic<int> cached{0};
  int v = cached.load();
  if (v > 0) return v;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
  v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  cached.store(v);
  return v;
}

template <int NM>
static int launch_indiv_mstep_t(int M, const int* order, const int64_t* off, const double* t,
                                const double* y,
                                const int* idx, const double* mean, const double* cpack, int mode,
                                const double* theta0, int lm, double factr, double pgtol,
                                int maxit, double* theta_out, double* f_out, double* fx0_out,
                                int* iters, int* nfev, int* status, hipStream_t st) {
  constexpr int RW = resident_waves<NM>();
  const size_t dyn = (size_t)RW * resident_wave_doubles<NM>() * sizeof(double);
  static AttrOnce once;
  if (int rc = once.ensure([&] {
        return check_hip(hipFuncSetAttribute((const void*)indiv_mstep_kernel<NM>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn),
                         "indiv_mstep LDS size");
      }))
    return rc;
  int* ticket = nullptr;
  if (int rc = resident_ticket(&ticket, st)) return rc;
  const int cap = t_resident_max_wgs >= 0 ? t_resident_max_wgs : g_resident_max_wgs.load(std::memory_order_relaxed);
  const int cus = device_compute_units();
  int grid = (M + RW - 1) / RW;
  if (grid > cus || (g_resident_help.load(std::memory_order_relaxed) & 2)) grid = cus;  // persistent: one workgroup fills a compute unit
  if (cap > 0 && grid > cap) grid = cap;
  const double* gsrc = t_resident_gather_cov;
  const int64_t gld = t_resident_gather_ld;
  t_resident_gather_cov = nullptr;  // (one launch only)
  hipLaunchKernelGGL(indiv_mstep_kernel<NM>, dim3(grid), dim3(64 * RW), dyn, st, M, order, off,
                     t, y, idx, mean, cpack, mode, theta0, lm, factr, pgtol, maxit, theta_out, f_out,
                     fx0_out, iters, nfev, status, ticket, g_resident_help.load(std::memory_order_relaxed) & 1,
                     g_resident_slow.load(std::memory_order_relaxed), gsrc, gld);
  return check_hip(hipGetLastError(), "indiv_mstep");
}

int launch_indiv_mstep(int M, int nmax, const int* order, const int64_t* off, const double* t,
                       const double* y,
                       const int* idx, const double* mean, const double* cpack, int mode,
                       const double* theta0, int lm, double factr, double pgtol, int maxit,
                       double* theta_out, double* f_out, double* fx0_out, int* iters, int* nfev,
                       int* status, hipStream_t st) {
  if (M <= 0) return 0;
  int rc = 0;
#define CALL(NM)                                                                               \
  rc = launch_indiv_mstep_t<NM>(M, order, off, t, y, idx, mean, cpack, mode, theta0, lm, factr, pgtol, \
                                maxit, theta_out, f_out, fx0_out, iters, nfev, status, st)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  return rc;
}

static std::atomic<bool> g_dedup{true};
bool dedup_enabled() { return g_dedup.load(); }
void set_dedup(bool on) { g_dedup.store(on); }
static std::atomic<bool> g_prefetch{true};
bool prefetch_enabled() { return g_prefetch.load(); }
void set_prefetch(bool on) { g_prefetch.store(on); }
static std::atomic<bool> g_covfac{true};
static std::atomic<bool> g_theta0_shard{true};
bool theta0_shard_enabled() { return g_theta0_shard.load(); }
void set_theta0_shard(bool on) { g_theta0_shard.store(on); }
bool covfac_enabled() { return g_covfac.load(); }
void set_covfac(bool on) { g_covfac.store(on); }
static std::atomic<bool> g_prefactor{true};
bool prefactor_enabled() { return g_prefactor.load(); }
void set_prefactor(bool on) { g_prefactor.store(on); }

int launch_dedup_prepare(int M, int n, int nmax, const double* y, const int* idx, const double* mean,
                         const double* cov, int64_t ldc, double* S, double* cpack, double* y_out,
                         hipStream_t st) {
  if (M < 1 || n < 1 || n > MAGMA_NSMALL_MAX) return fail(2, "dedup: bad sizes");
  const int nc = small_template_nmax(nmax), nt = (nc + 15) / 16;
  hipLaunchKernelGGL(dedup_outer_kernel, dim3((unsigned)n), dim3(64 * DEDUP_PARTS), 0, st, M, n, y, idx, mean, S);
  hipLaunchKernelGGL(dedup_pack_kernel, dim3(1), dim3(256), 0, st, n, idx, cov, ldc, S, mean, cpack,
                     y_out, nt, nc);
  return check_hip(hipGetLastError(), "dedup_prepare");
}

int launch_indiv_gather_cov(int M, int nmax, const int64_t* off, const int* idx,
                            const double* cov, int64_t ldc, double* cpack, hipStream_t st) {
  if (M <= 0) return 0;
  const int grid = M < 65536 ? M : 65536;
  hipLaunchKernelGGL(indiv_gather_cov_kernel, dim3(grid), dim3(256), 0, st, M, off, idx, cov, ldc,
                     cpack, cpack_ldp(nmax) / 16, small_template_nmax(nmax));
  return check_hip(hipGetLastError(), "indiv_gather_cov");
}

int launch_pred_gp(int n, const double* tn, const double* yn, const double* mean_obs,
                   const int* idx_n, int T, const double* ts, const double* mean_pred,
                   const int* idx_t, const double* cov, int64_t ldc, const double* theta,
                   double* out_mean, double* out_var, int* info, hipStream_t st) {
  if (T <= 0) return 0;
  const int chunk = 8;
  const int grid = (T + chunk - 1) / chunk;
#define CALL(NM)                                                                            \
  hipLaunchKernelGGL(pred_gp_kernel<NM>, dim3(grid), dim3(64),                              \
                     (NM * NM + 16) * sizeof(double), st, n, tn, yn, mean_obs,               \
                     idx_n, T, ts, mean_pred, idx_t, cov, ldc, theta, out_mean, out_var,   \
                     info, chunk)
  MAGMA_DISPATCH_NMAX(n, CALL);
#undef CALL
  if (int rc = check_hip(hipGetLastError(), "pred_gp")) return rc;
  const size_t dyn = (size_t)SP_LDS_DOUBLES * sizeof(double);
  if (int rc = slowpath_lds_attr((const void*)pred_gp_slowpath_kernel, dyn, 2)) return rc;
  hipLaunchKernelGGL(pred_gp_slowpath_kernel, dim3(1), dim3(64), dyn, st, n, tn, yn, mean_obs, idx_n, T,
                     ts, mean_pred, idx_t, cov, ldc, theta, out_mean, out_var, info);
  return check_hip(hipGetLastError(), "pred_gp slow path");
}

int launch_pred_gp_batched(int B, int nmax, int tmax, const int64_t* off_n, const double* tn,
                           const double* yn, const int* idx_n, const int64_t* off_t,
                           const double* ts, const int* idx_t, const double* mean_grid,
                           const double* cov, int64_t ldc, const double* theta, int tstride,
                           double* out_mean, double* out_var, int* info, hipStream_t st) {
  if (B <= 0) return 0;
  const int chunk = 8;
  const int gx = tmax > 0 ? (tmax + chunk - 1) / chunk : 1;
#define CALL(NM)                                                                               \
  hipLaunchKernelGGL(pred_gp_batched_kernel<NM>, dim3(gx, B), dim3(64),                        \
                     (NM * NM + 16) * sizeof(double), st, off_n, tn, yn,                        \
                     idx_n, off_t, ts, idx_t, mean_grid, cov, ldc, theta, tstride, out_mean,   \
                     out_var, info, chunk)
  MAGMA_DISPATCH_NMAX(nmax, CALL);
#undef CALL
  if (int rc = check_hip(hipGetLastError(), "pred_gp_batched")) return rc;
  const size_t dyn = (size_t)SP_LDS_DOUBLES * sizeof(double);
  if (int rc = slowpath_lds_attr((const void*)pred_gp_batched_slowpath_kernel, dyn, 3)) return rc;
  hipLaunchKernelGGL(pred_gp_batched_slowpath_kernel, dim3(B < 512 ? B : 512), dim3(64), dyn, st, B,
                     off_n, tn, yn, idx_n, off_t, ts, idx_t, mean_grid, cov, ldc, theta, tstride,
                     out_mean, out_var, info);
  return check_hip(hipGetLastError(), "pred_gp_batched slow path");
}

}  // namespace magma

#ifdef MAGMA_STAMPS
extern "C" int magma_dbg_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(magma::g_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(magma::g_stamps), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
