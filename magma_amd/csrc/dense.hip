// Dense union-grid kernels (N x N, FP64): SE covariance on the grid, blocked right-looking
// Cholesky, triangular inverse by recursive doubling, A^-1 = L^-T L^-1, symmetric products
// and the theta_0 gradient contraction -- all on v_mfma_f64_16x16x4_f64 tiles.
//
// Conventions: every matrix is Np x Np, Np a multiple of 64, column-major with ld = Np; rows
// and columns >= N are padding that carries the identity (so inverse, log-det and products
// are unaffected). Tiles are 64 x 64, one 256-thread workgroup (4 wavefronts, 2 x 2, each a
// 32 x 32 sub-tile = 2 x 2 MFMA blocks). Operands are staged through LDS as S[k][x]
// (row length 80 doubles: the four 16-lane groups of one MFMA operand read hit disjoint banks).
//
// Replaces (Computing_functions.R = CF): kern_to_cov/kern_to_inv vector branch CF:66-81,
// 121-138; solve(new_inv) CF:387; logL_GP_mod / gr_GP_mod at theta_0 (CF:181-201, 304-332).
#include "common.h"
#include <atomic>
#include <algorithm>
#include "kernels.h"
#include "mfma_sweep.h"
#include "dense_tile.h"
#include "driver.h"

namespace magma {

// acc(m, n) += sum_k Aop(m, k) * Bop(n, k), Aop(m,k) = Ap[m + k*lda], Bop(n,k) = Bp[n + k*ldb],
// m, n in [0,64), k in [0, 16*ksteps). As, Bs: 16*TLD doubles of LDS each.
// MFMA operand roles are swapped (first operand = B rows) so that the accumulator has m on the
// fast lane index: lane l, register r holds C(m = wm*32 + mi*16 + (l & 15),
// n = wn*32 + ni*16 + (l >> 4) + 4 r) -- column-major stores are then 128-byte segments.
__device__ __forceinline__ void tile_gemm_nt(const double* __restrict__ Ap, int64_t lda,
                                             const double* __restrict__ Bp, int64_t ldb,
                                             int ksteps, TileAcc& acc, double* As, double* Bs) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int x4 = (tid & 15) * 4, kr = tid >> 4;
  const double2* ga = reinterpret_cast<const double2*>(Ap + x4 + (int64_t)kr * lda);
  const double2* gb = reinterpret_cast<const double2*>(Bp + x4 + (int64_t)kr * ldb);
  double2 ra0 = ga[0], ra1 = ga[1], rb0 = gb[0], rb1 = gb[1];
  const int fo = (lane >> 4) * TLD + (lane & 15);
  for (int s = 0; s < ksteps; ++s) {
    __syncthreads();
    *reinterpret_cast<double2*>(&As[kr * TLD + x4]) = ra0;
    *reinterpret_cast<double2*>(&As[kr * TLD + x4 + 2]) = ra1;
    *reinterpret_cast<double2*>(&Bs[kr * TLD + x4]) = rb0;
    *reinterpret_cast<double2*>(&Bs[kr * TLD + x4 + 2]) = rb1;
    __syncthreads();
    if (s + 1 < ksteps) {
      ga = reinterpret_cast<const double2*>(Ap + x4 + (int64_t)(kr + 16 * (s + 1)) * lda);
      gb = reinterpret_cast<const double2*>(Bp + x4 + (int64_t)(kr + 16 * (s + 1)) * ldb);
      ra0 = ga[0];
      ra1 = ga[1];
      rb0 = gb[0];
      rb1 = gb[1];
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      const double a0 = As[kk * TLD + fo + wm * 32];
      const double a1 = As[kk * TLD + fo + wm * 32 + 16];
      const double b0 = Bs[kk * TLD + fo + wn * 32];
      const double b1 = Bs[kk * TLD + fo + wn * 32 + 16];
      acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc.c[0][0], 0, 0, 0);
      acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc.c[0][1], 0, 0, 0);
      acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc.c[1][0], 0, 0, 0);
      acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc.c[1][1], 0, 0, 0);
    }
  }
}

// ------------------------------------------------------------------ K2: SE covariance on the grid
__global__ __launch_bounds__(256) void se_cov_grid_kernel(int N, int Np,
                                                          const double* __restrict__ grid,
                                                          double th1, double th2, double sigma,
                                                          double* __restrict__ A) {
  __shared__ double gr[64], gc[64];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  if (tid < 64) gr[tid] = (r0 + tid < N) ? grid[r0 + tid] : 0.0;
  else if (tid < 128) gc[tid - 64] = (c0 + tid - 64 < N) ? grid[c0 + tid - 64] : 0.0;
  __syncthreads();
  const SEParams p = se_params(th1, th2, sigma);
  for (int e = tid; e < 4096; e += 256) {
    const int r = e & 63, c = e >> 6;
    const int gi = r0 + r, gj = c0 + c;
    double v;
    if (gi < N && gj < N) {
      const double d = gr[r] - gc[c];
      v = (gi == gj) ? (p.eth1 + p.sig2) : exp(p.th1 - p.h * d * d);
    } else {
      v = (gi == gj) ? 1.0 : 0.0;
    }
    A[(int64_t)gj * Np + gi] = v;
  }
}

// ------------------------------------------------------------------ Cholesky panel
// Step k of the right-looking factorisation. Every workgroup re-factors the 64 x 64 diagonal
// block A_kk in registers/LDS (cheaper than a launch boundary), then workgroup w > 0 solves
// L_ik = A_ik L_kk^-T for block row i = k + w, and workgroup 0 stores L_kk and solves the
// same system against the identity, which yields U_kk = L_kk^-T and X_kk = L_kk^-1 (the
// leaves of the triangular inverse).
// Thread map: row = 16 * wave + (lane >> 2), ty = lane & 3; a thread owns the 16 elements
// [row][4q + ty]. A row lives in one quad, so "my row's value of column j" is a DPP quad
// broadcast; only the factorisation needs LDS + a barrier per column (other rows' values).
// Reciprocals / reciprocal square roots are v_rcp/v_rsq + Newton steps: no IEEE division or
// sqrt on the per-column dependency chain.
template <int SRC>
__device__ __forceinline__ double quad_bcast_t(double v) {
  constexpr int ctrl = SRC * 0x55;  // quad_perm: every lane of the quad reads lane SRC
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double quad_bcast(double v, int src) {
  switch (src & 3) {
    case 0: return quad_bcast_t<0>(v);
    case 1: return quad_bcast_t<1>(v);
    case 2: return quad_bcast_t<2>(v);
    default: return quad_bcast_t<3>(v);
  }
}

__device__ __forceinline__ double fast_rcp_d(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}

__device__ __forceinline__ double fast_rsqrt_d(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  return y;
}

// Factor the 64 x 64 SPD diagonal block whose elements the workgroup holds as d[16] per thread
// (thread map above) and emit X_kk = L_kk^-1, U_kk = L_kk^-T (the leaves of the triangular
// inverse, and the operand that turns the block-row solve into an MFMA product), log-det and
// info. LDS: Ls[128 * 65] (rows >= 64 are only read into dead registers), colbuf[2][128]
// (entries >= 64 likewise), piv[64], invd[64].
// The column loops run over 16 quads of 4 columns with the register window ROTATED after each
// quad (d[0] is always the quad being eliminated), so the body is a few KB that stays in the
// instruction cache (a fully unrolled 64-column body is ~70 KB of straight-line code and ran
// instruction-fetch bound). Each wavefront issues the whole column step on its own SIMD, so the
// per-column instruction count is what sets the time: the loop keeps only the rank-1 update;
// scaling by 1/L_jj, the log-det and the SPD check are vector passes after it.
__device__ __forceinline__ void potrf_diag_block(double (&d)[16], double* Ls, double* colbuf,
                                                 double* piv, double* invd, int k, int64_t ld,
                                                 double* __restrict__ X, double* __restrict__ U,
                                                 double* __restrict__ logdet,
                                                 int* __restrict__ info) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = 16 * wave + (lane >> 2), ty = lane & 3;
  const int ldi = (int)ld;  // block-local offsets fit 32 bits (64 columns of one block)
  double b[16];  // identity: X L^T = I yields L_kk^-1
#pragma unroll
  for (int q = 0; q < 16; ++q) b[q] = (4 * q + ty == row) ? 1.0 : 0.0;
  if (tid < 128) {
    colbuf[tid] = 0.0;
    colbuf[128 + tid] = 0.0;
  }
  for (int e = tid; e < 64 * 65; e += 256) Ls[64 * 65 + e] = 0.0;
  __syncthreads();
#pragma unroll 1
  for (int q0 = 0; q0 < 16; ++q0) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * q0 + jj;
      double* cb = colbuf + 128 * (jj & 1);
      if (ty == jj) {
        cb[row] = d[0];
        Ls[row * 65 + j] = d[0];  // unscaled column j (rows >= j are final); scaled below
      }
      __syncthreads();
      const double djj = cb[j];
      const double own = quad_bcast(d[0], jj);
      const double f = own * fast_rcp_d(djj);
      // Branch-free rank-1 update of every column right of j. Elements above the diagonal
      // (and the dead registers behind column 63) are updated too, with values that are
      // never read, so that no per-element predicate is needed.
      const double f0 = (ty > jj) ? f : 0.0;
      const double* cq = cb + 4 * q0 + ty;
      d[0] = fma(-f0, cq[0], d[0]);
#pragma unroll
      for (int q = 1; q < 16; ++q) d[q] = fma(-f, cq[4 * q], d[q]);
      if (tid == 0) piv[j] = djj;
    }
#pragma unroll
    for (int q = 0; q < 15; ++q) d[q] = d[q + 1];
    d[15] = 0.0;
  }
  __syncthreads();
  // pivots -> 1/L_jj, SPD check and log-det (first wavefront); then L = columns * 1/L_jj
  if (wave == 0) {
    const double pj = piv[lane];
    const bool ok = (pj > 0.0) && (pj < 1.7e308);
    invd[lane] = fast_rsqrt_d(ok ? pj : 1.0);
    const unsigned long long badmask = __ballot(!ok);
    int e;
    const double m = frexp(ok ? pj : 1.0, &e);
    double lg = log(m) + (double)e * 0.69314718055994530941723212145818;
    lg = wave_sum(lg);
    if (lane == 0) {
      logdet[0] += lg;  // one workgroup per step, steps are stream ordered
      if (badmask != 0ull && info[0] == 0) info[0] = k * NB + __ffsll((long long)badmask);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = 4 * q + ty;
    const double v = Ls[row * 65 + c] * invd[c];
    Ls[row * 65 + c] = (row >= c) ? v : 0.0;
  }
  __syncthreads();
  // X L^T = I, column by column; rows are independent and each row sits in one quad, so this
  // phase needs no barrier. b[0] is final once its quad is done and is stored right away.
  double* Ukk = U + (int64_t)k * NB * (ld + 1);
  double* Xkk = X + (int64_t)k * NB * (ld + 1);
#pragma unroll 1
  for (int q0 = 0; q0 < 16; ++q0) {
    const double* lq = Ls + (4 * q0 + ty) * 65 + 4 * q0;  // L[4 q0 + ty][4 q0]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const double xo = b[0] * invd[4 * q0 + jj];
      b[0] = (ty == jj) ? xo : b[0];
      const double x = quad_bcast(xo, jj);
      const double x0 = (ty > jj) ? x : 0.0;
      b[0] = fma(-x0, lq[jj], b[0]);
#pragma unroll
      for (int q = 1; q < 16; ++q) b[q] = fma(-x, lq[q * 4 * 65 + jj], b[q]);
    }
    const int c = 4 * q0 + ty;
    Ukk[row + c * ldi] = b[0];
    Xkk[c + row * ldi] = b[0];
#pragma unroll
    for (int q = 0; q < 15; ++q) b[q] = b[q + 1];
    b[15] = 0.0;
  }
}

// The same job on the matrix cores, by two wavefronts of the (256-thread) workgroup: blocked
// Cholesky of the staged tile S (S[m * 65 + n], lower triangle valid) in wavefront 0, the
// identity carried along in wavefront 1 (chol64_factor / chol64_solve): 200 MFMAs + 16 4x4
// factorisations instead of 64 barrier-separated column steps and a 64-step triangular solve.
// Called by ALL wavefronts of the workgroup. LDS: Pn[512] (Pn, Bn), Xch[CHOL64_XCH].
__device__ __forceinline__ void potrf_diag_block_mfma(const double* S, double* Pn, double* Xch,
                                                      int k, int64_t ld, double* __restrict__ X,
                                                      double* __restrict__ U,
                                                      double* __restrict__ logdet,
                                                      int* __restrict__ info) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane & 15, g = lane >> 4;
  if (wave == 0) {
    v4d A[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * it + g + 4 * v, c = 16 * jt + l;
          A[it][jt][v] = (r >= c) ? S[r * 65 + c] : S[c * 65 + r];
        }
    chol64_factor(A, Pn, Xch, lane);
  } else if (wave == 1) {
    v4d B[4][4];
    double pm = 1.0;
    int pe = 0, bad = 0;
    chol64_solve(B, Pn + 256, Xch, lane, pm, pe, bad);
    if (lane == 0) {
      logdet[0] += log(pm) + (double)pe * 0.69314718055994530941723212145818;
      if (bad != 0 && info[0] == 0) info[0] = k * NB + bad;
    }
    double* Ukk = U + (int64_t)k * NB * (ld + 1);
    double* Xkk = X + (int64_t)k * NB * (ld + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * it + g + 4 * v, c = 16 * jt + l;
          const double x = (jt <= it) ? B[it][jt][v] : 0.0;  // X = L^-1 is lower triangular
          Xkk[r + (int64_t)c * ld] = x;
          Ukk[c + (int64_t)r * ld] = x;
        }
  } else {
    chol64_idle();
  }
}

// Step 0: the first diagonal block (one workgroup).
__global__ __launch_bounds__(256) void potrf_diag0_kernel(const double* __restrict__ A, int64_t ld,
                                                          double* __restrict__ X,
                                                          double* __restrict__ U,
                                                          double* __restrict__ logdet,
                                                          int* __restrict__ info) {
#ifdef MAGMA_DIAG_VECTOR
  __shared__ double Ls[128 * 65];
  __shared__ double colbuf[2 * 128];
  __shared__ double piv[64];
  __shared__ double invd[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = 16 * wave + (lane >> 2), ty = lane & 3;
  double d[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) d[q] = A[row + (int64_t)(4 * q + ty) * ld];
  potrf_diag_block(d, Ls, colbuf, piv, invd, 0, ld, X, U, logdet, info);
#else
  __shared__ double Ls[64 * 65];
  __shared__ __attribute__((aligned(16))) double Pn[512];
  __shared__ __attribute__((aligned(16))) double Xch[CHOL64_XCH];
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int m = e & 63, n = e >> 6;
    Ls[m * 65 + n] = A[m + (int64_t)n * ld];
  }
  __syncthreads();
  potrf_diag_block_mfma(Ls, Pn, Xch, 0, ld, X, U, logdet, info);
#endif
}

// Block-row solve of step k as an MFMA product: L_ik = A_ik L_kk^-T = A_ik X_kk^T, in place.
__global__ __launch_bounds__(256) void trsm_gemm_kernel(double* __restrict__ A, int64_t ld, int k,
                                                        const double* __restrict__ X) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  const int i = k + 1 + blockIdx.x;
  double* Aik = A + (int64_t)i * NB + (int64_t)k * NB * ld;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(Aik, ld, X + (int64_t)k * NB * (ld + 1), ld, 4, acc, As, Bs);
  __syncthreads();  // every wavefront has consumed A_ik before it is overwritten
  tile_foreach(acc, [&](int m, int n, double v) { Aik[m + (int64_t)n * ld] = v; });
}

// ------------------------------------------------------------------ trailing update
// A_ij -= sum over the kc finished panels k .. k + kc - 1 of L_i. L_j.^T (the panels are adjacent
// block columns: one product with K = 64 kc), for the target tiles
//   col_only = 0: j0 <= j <= i  (everything right of the panels; j0 = k + kc),
//   col_only = 1: j = j0 <= i   (block column j0 only: the next panel of a super-step).
// A rank-64 kc update reads and writes each target tile once per kc panels instead of once per
// panel: at N = 8,000 the rank-64 form spends its time moving the trailing matrix (4 flops per
// byte), which is what the super-steps of launch_spd_inverse cut. The workgroup of tile (j0, j0)
// -- block 0, dispatched first -- goes on to factor that tile and emit X / U of step j0
// (potrf_diag_block) while the other workgroups finish the update: the serial part of the next
// step hides behind this one.
__global__ __launch_bounds__(256) void syrk_update_kernel(double* __restrict__ A, int64_t ld,
                                                          int k, int kc, int j0, int col_only,
                                                          double* __restrict__ X,
                                                          double* __restrict__ U,
                                                          double* __restrict__ logdet,
                                                          int* __restrict__ info) {
#ifdef MAGMA_DIAG_VECTOR
  __shared__ double Ls[128 * 65];
  __shared__ double colbuf[2 * 128];
  __shared__ double piv[64];
  __shared__ double invd[64];
  double* As = Ls + 64 * 65;  // operand slabs in the half of Ls that is dead space until
  double* Bs = As + 16 * TLD;  // potrf_diag_block zeroes it
#else
  __shared__ __attribute__((aligned(16))) double Ls[64 * 65];  // staged diagonal tile; its head
  __shared__ __attribute__((aligned(16))) double Pn[512];      // doubles as the operand slabs
  __shared__ __attribute__((aligned(16))) double Xch[CHOL64_XCH];
  static_assert(2 * 16 * TLD <= 64 * 65, "operand slabs must fit the staging tile");
  double* As = Ls;
  double* Bs = As + 16 * TLD;
#endif
  // lower-triangle tile index -> (bi, bj), bj <= bi; tile 0 = (0, 0) = block (j0, j0)
  const int p = blockIdx.x;
  int bi = p, bj = 0;
  if (!col_only) {
    bi = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (bi * (bi + 1) / 2 > p) --bi;
    while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
    bj = p - bi * (bi + 1) / 2;
  }
  const int i = j0 + bi, j = j0 + bj;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(A + (int64_t)i * NB + (int64_t)k * NB * ld, ld,
               A + (int64_t)j * NB + (int64_t)k * NB * ld, ld, 4 * kc, acc, As, Bs);
  double* C = A + (int64_t)i * NB + (int64_t)j * NB * ld;
  if (p != 0) {
    tile_foreach(acc, [&](int m, int n, double v) { C[m + (int64_t)n * ld] -= v; });
    return;
  }
  // diagonal block of the next step: updated tile -> LDS -> the factorisation's thread map
  // (not written back: nothing reads L_kk or A_kk afterwards)
  __syncthreads();
  tile_foreach(acc, [&](int m, int n, double v) { Ls[m * 65 + n] = C[m + (int64_t)n * ld] - v; });
  __syncthreads();
#ifdef MAGMA_DIAG_VECTOR
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = 16 * wave + (lane >> 2), ty = lane & 3;
  double d[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) d[q] = Ls[row * 65 + 4 * q + ty];
  __syncthreads();
  potrf_diag_block(d, Ls, colbuf, piv, invd, j0, ld, X, U, logdet, info);
#else
  potrf_diag_block_mfma(Ls, Pn, Xch, j0, ld, X, U, logdet, info);
#endif
}

// ------------------------------------------------------------------ triangular inverse
// Node p at `depth` of the balanced binary tree over block indices [0, nt).
__device__ __forceinline__ void node_range(int nt, int depth, int p, int& lo, int& mid,
                                           int& hi) {
  lo = 0;
  hi = nt;
  for (int b = depth - 1; b >= 0; --b) {
    const int m = (lo + hi) >> 1;
    if ((p >> b) & 1) lo = m; else hi = m;
  }
  mid = (lo + hi) >> 1;
}

// Step A: W[b1, b2] = U11 L21^T  (U upper triangular: k-tiles start at the m-tile).
__global__ __launch_bounds__(256) void trtri_a_kernel(const double* __restrict__ L,
                                                      const double* __restrict__ U,
                                                      double* __restrict__ W, int64_t ld,
                                                      int nt, int depth) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  int lo, mid, hi;
  node_range(nt, depth, blockIdx.z, lo, mid, hi);
  if (hi - lo < 2) return;
  const int mt = lo + blockIdx.x, ntile = mid + blockIdx.y;
  if (mt >= mid || ntile >= hi) return;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(U + (int64_t)mt * NB + (int64_t)mt * NB * ld, ld,
               L + (int64_t)ntile * NB + (int64_t)mt * NB * ld, ld, (mid - mt) * 4, acc, As, Bs);
  double* C = W + (int64_t)mt * NB + (int64_t)ntile * NB * ld;
  tile_foreach(acc, [&](int m, int n, double v) { C[m + (int64_t)n * ld] = v; });
}

// Step B: U12 = -W U22 = -W X22^T (X lower triangular: k-tiles end at the n-tile); X21 = U12^T.
__global__ __launch_bounds__(256) void trtri_b_kernel(const double* __restrict__ W,
                                                      double* __restrict__ X,
                                                      double* __restrict__ U, int64_t ld, int nt,
                                                      int depth) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  int lo, mid, hi;
  node_range(nt, depth, blockIdx.z, lo, mid, hi);
  if (hi - lo < 2) return;
  const int mt = lo + blockIdx.x, ntile = mid + blockIdx.y;
  if (mt >= mid || ntile >= hi) return;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(W + (int64_t)mt * NB + (int64_t)mid * NB * ld, ld,
               X + (int64_t)ntile * NB + (int64_t)mid * NB * ld, ld, (ntile - mid + 1) * 4, acc,
               As, Bs);
  double* Cu = U + (int64_t)mt * NB + (int64_t)ntile * NB * ld;
  double* Cx = X + (int64_t)ntile * NB + (int64_t)mt * NB * ld;
  tile_foreach(acc, [&](int m, int n, double v) {
    Cu[m + (int64_t)n * ld] = -v;
    Cx[n + (int64_t)m * ld] = -v;
  });
}

// A^-1 = U U^T: tile (i >= j) = sum_{k >= i} U(i-block, k) U(j-block, k)^T, mirrored.
__global__ __launch_bounds__(256) void lauum_kernel(const double* __restrict__ U,
                                                    double* __restrict__ Ainv, int64_t ld,
                                                    int nt) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  const int i = blockIdx.x, j = blockIdx.y;
  if (j > i) return;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(U + (int64_t)i * NB + (int64_t)i * NB * ld, ld,
               U + (int64_t)j * NB + (int64_t)i * NB * ld, ld, (nt - i) * 4, acc, As, Bs);
  double* C = Ainv + (int64_t)i * NB + (int64_t)j * NB * ld;
  double* Ct = Ainv + (int64_t)j * NB + (int64_t)i * NB * ld;
  const bool diag = (i == j);
  tile_foreach(acc, [&](int m, int n, double v) {
    C[m + (int64_t)n * ld] = v;
    if (!diag) Ct[n + (int64_t)m * ld] = v;
  });
}

// ------------------------------------------------------------------ general C = alpha A B^T + beta C
__global__ __launch_bounds__(256, 8) void gemm_nt_kernel(int K, double alpha,
                                                      const double* __restrict__ A, int64_t lda,
                                                      const double* __restrict__ B, int64_t ldb,
                                                      double beta, double* __restrict__ C,
                                                      int64_t ldc, int row0, int rstride,
                                                      int ktri) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  TileAcc acc;
  tile_zero(acc);
  // block row of this workgroup: all of them (row0 = 0, rstride = 1) or one rank's share
  const int64_t bi = row0 + (int64_t)rstride * blockIdx.x;
  // ktri: B is lower triangular (B(n, k) = 0 for k > n): column tile bj sums k < 64 (bj + 1) only;
  // the column tiles are then handed out longest first
  const int bj = ktri ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
  const int kend = ktri ? min(K, NB * (bj + 1)) : K;
  tile_gemm_nt(A + bi * NB, lda, B + (int64_t)bj * NB, ldb, kend / 16, acc, As, Bs);
  double* Ct = C + bi * NB + (int64_t)bj * NB * ldc;
  if (beta == 0.0) {
    tile_foreach(acc, [&](int m, int n, double v) { Ct[m + (int64_t)n * ldc] = alpha * v; });
  } else {
    tile_foreach(acc, [&](int m, int n, double v) {
      double* p = &Ct[m + (int64_t)n * ldc];
      *p = alpha * v + beta * *p;
    });
  }
}

// ------------------------------------------------------------------ theta_0 gradient contraction
// W = alpha alpha' + Z - Kinv with Z = T Kinv, contracted with K_nl and dK/dtheta2 over the true
// N x N region; lower tiles only (off-diagonal tiles x2). Two launches: the product of the lower
// tiles into a scratch matrix (16 MB at N = 2000: a few microseconds of HBM time) and an
// element-wise pass. Both are held to 64 registers per lane (8 waves per SIMD), like
// gemm_nt_kernel: the theta_0 chain runs next to the device-resident M-step, whose wavefronts
// hold 448 registers each; a 64-register wavefront still fits beside one on the same SIMD and
// uses the MFMA cycles it leaves idle instead of waiting for a whole SIMD to retire (a fused
// product + epilogue kernel needs 108 registers and took 1.0 ms there against 0.2 ms alone).
// k-slices: slice z of ks sums k in [z K / ks, (z + 1) K / ks) into its own
// copy of C (C + z * cslice). Half the tiles are skipped, so at N = 2000 one slice per tile leaves
// two 64-register workgroups per CU, too few to hide their own LDS and barrier latencies
// (0.27 ms for N^3 flops against 0.28 ms for the full product's 2 N^3); two slices fill the chip.
__global__ __launch_bounds__(256, 8) void gemm_nt_lower_kernel(int K, const double* __restrict__ A,
                                                               int64_t lda,
                                                               const double* __restrict__ B,
                                                               int64_t ldb, double* __restrict__ C,
                                                               int64_t ldc, int64_t cslice,
                                                               int ks,
                                                               const int* __restrict__ plist) {
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  // 1-D grid over (lower tile, slice): consecutive workgroups go to different XCDs, so every XCD
  // gets the same number of (equal-sized) tasks; a 2-D grid with the upper half skipped gives
  // XCD x the block rows i = x mod 8, 52 to 80 tiles each.
  const int z = (int)blockIdx.x % ks;
  const int p = plist ? plist[(int)blockIdx.x / ks] : (int)blockIdx.x / ks;  // (a rank's tiles)
  int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > p) --i;
  while ((i + 1) * (i + 2) / 2 <= p) ++i;
  const int j = p - i * (i + 1) / 2;
  const int ksl = K / ks;  // multiple of 16 (checked by the launcher)
  const int64_t k0 = (int64_t)z * ksl;
  TileAcc acc;
  tile_zero(acc);
  tile_gemm_nt(A + (int64_t)i * NB + k0 * lda, lda, B + (int64_t)j * NB + k0 * ldb, ldb, ksl / 16,
               acc, As, Bs);
  double* Ct = C + (int64_t)z * cslice + (int64_t)i * NB + (int64_t)j * NB * ldc;
  tile_foreach(acc, [&](int m, int n, double v) { Ct[m + (int64_t)n * ldc] = v; });
}

__global__ __launch_bounds__(256, 8) void theta0_contract_kernel(
    int N, int Np, const double* __restrict__ Z, int zslices, int64_t zslice,
    const double* __restrict__ Kinv, const double* __restrict__ alpha,
    const double* __restrict__ grid, double th1, double th2, double* __restrict__ partials,
    int rank, int world) {
  __shared__ double red[2][4];
  const int i = blockIdx.x, j = blockIdx.y, nt = gridDim.x;
  double* out = partials + 2 * ((int64_t)j * nt + i);
  if (j > i || (i % world) != rank) {  // upper tiles by symmetry; other ranks' block rows
    if (threadIdx.x == 0) {
      out[0] = 0.0;
      out[1] = 0.0;
    }
    return;
  }
  const int64_t ld = Np;
  const double h = 0.5 * exp(-th2);
  const double eth1 = exp(th1);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gm = i * NB + lane;
  const double am = alpha[gm], tm = grid[gm];
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int gn = j * NB + wave + 4 * r;
    if (gm < N && gn < N) {
      const int64_t o = gm + (int64_t)gn * ld;
      double z = Z[o];
      for (int q = 1; q < zslices; ++q) z += Z[o + q * zslice];
      const double W = z + am * alpha[gn] - Kinv[o];
      const double d = tm - grid[gn];
      const double gq = h * d * d;
      const double knl = (gm == gn) ? eth1 : exp(th1 - gq);
      s1 += W * knl;
      s2 += W * knl * gq;
    }
  }
  const double wgt = (i == j) ? 1.0 : 2.0;
  s1 = wave_sum(s1) * wgt;
  s2 = wave_sum(s2) * wgt;
  if (lane == 0) {
    red[0][wave] = s1;
    red[1][wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

__global__ __launch_bounds__(256) void reduce_pairs_kernel(int64_t n, const double* __restrict__ p,
                                                           double* __restrict__ out2) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    s1 += p[2 * e];
    s2 += p[2 * e + 1];
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wave] = s1;
    red[1][wave] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out2[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    out2[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// ------------------------------------------------------------------ theta_0 objective, vector part
// One pass over K_0^-1 for everything the objective needs from it besides the log-det:
// alpha = K_0^-1 z with z = mean - m0 on the true grid (zero on the padding), z'alpha and
// sum(K_0^-1 o cov) as per-workgroup partial sums (vpart[2 b], vpart[2 b + 1]); one wavefront per
// column. Replaces a copy, an axpy, a memset, a symv and two two-stage dot products.
__global__ __launch_bounds__(256) void theta0_vec_kernel(int N, int Np,
                                                         const double* __restrict__ Kinv,
                                                         const double* __restrict__ cov,
                                                         const double* __restrict__ mean,
                                                         const double* __restrict__ m0,
                                                         double* __restrict__ alpha,
                                                         double* __restrict__ vpart) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  double s = 0.0, c = 0.0;
  if (j < Np) {
    const double* col = Kinv + (int64_t)j * Np;
    const double* ccol = cov + (int64_t)j * Np;
    for (int i = lane; i < Np; i += 64) {
      const double k = col[i];
      const double zi = (i < N) ? mean[i] - m0[i] : 0.0;
      s = fma(k, zi, s);
      c = fma(k, ccol[i], c);
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if (lane == 0) alpha[j] = s;
    s *= (j < N) ? mean[j] - m0[j] : 0.0;
  }
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    vpart[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    vpart[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// out4 = (f, g1, g2, info) from the log-det, the vector part's partial sums and (with want_grad)
// the contraction's per-tile pairs; one workgroup, fixed summation order.
__global__ __launch_bounds__(256) void theta0_finish2_kernel(
    int N, int Np, const double* __restrict__ logdet, const int* __restrict__ info, int nvb,
    const double* __restrict__ vpart, int64_t npairs, const double* __restrict__ pairs,
    int want_grad, double* __restrict__ out4) {
  __shared__ double red[4][4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < nvb; e += 256) {
    v[0] += vpart[2 * e];
    v[1] += vpart[2 * e + 1];
  }
  if (want_grad) {
    for (int64_t e = threadIdx.x; e < npairs; e += 256) {
      v[2] += pairs[2 * e];
      v[3] += pairs[2 * e + 1];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = wave_sum(v[q]);
    if (lane == 0) red[q][wave] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
    // t[1] was summed over the padded Np x Np arrays: each pad diagonal entry adds 1 * 1
    out4[0] = 0.5 * ((double)N * 1.8378770664093454835606594728112 + logdet[0] + t[0]) +
              0.5 * (t[1] - (double)(Np - N));
    out4[1] = want_grad ? -0.5 * t[2] : 0.0;
    out4[2] = want_grad ? -0.5 * t[3] : 0.0;
    out4[3] = (double)info[0];
  }
}

// ------------------------------------------------------------------ BLAS-1/2 helpers
// y_j = sum_i A[i + j*ld] x_i  (A symmetric: column j == row j). One wavefront per column.
__global__ __launch_bounds__(256) void symv_kernel(int Np, const double* __restrict__ A,
                                                   const double* __restrict__ x,
                                                   double* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= Np) return;
  const double* col = A + (int64_t)j * Np;
  double s = 0.0;
  for (int i = lane; i < Np; i += 64) s = fma(col[i], x[i], s);
  s = wave_sum(s);
  if (lane == 0) y[j] = s;
}

__global__ void axpy_kernel(int64_t n, double alpha, const double* __restrict__ x,
                            double* __restrict__ y) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    y[e] = fma(alpha, x[e], y[e]);
}

__global__ void fill_kernel(int64_t n, double v, double* __restrict__ x) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n;
       e += (int64_t)gridDim.x * blockDim.x)
    x[e] = v;
}

// Test aid (magma_debug_poison_lds): every word of the workgroup's LDS becomes a NaN pattern.
__global__ __launch_bounds__(256) void poison_lds_kernel(int ndoubles, double* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) double pl[];
  const double nanv = __longlong_as_double(0x7ff8dead7ff8deadLL);
  for (int e = threadIdx.x; e < ndoubles; e += 256) pl[e] = nanv;
  __syncthreads();
  // keep the stores alive, and the workgroup on its CU long enough for the grid to spread
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0 && pl[(blockIdx.x * 7) % ndoubles] == 0.0) sink[0] = 1.0;
}

// out[0] = sum_e a[e] b[e] over n elements (deterministic two-stage: per-block partials in
// `scratch`, then one block). Used for z'alpha (vectors) and sum(Kinv o C) (N x N region).
__global__ __launch_bounds__(256) void dot_stage1_kernel(int64_t n, const double* __restrict__ a,
                                                         const double* __restrict__ b,
                                                         double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  if (b != nullptr) {
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
      s = fma(a[e], b[e], s);
  } else {  // plain sum
    for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256)
      s += a[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sum_stage2_kernel(int n, const double* __restrict__ partial,
                                                         double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int e = threadIdx.x; e < n; e += 256) s += partial[e];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// dst (Np x Np, ld Np) <- src (N x N, ld lds); padding: identity (zero_pad = 0) or zeros.
__global__ void copy_pad_kernel(int N, int Np, const double* __restrict__ src, int64_t lds,
                                double* __restrict__ dst, int zero_pad) {
  const int64_t total = (int64_t)Np * Np;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % Np), c = (int)(e / Np);
    double v;
    if (r < N && c < N) v = src[r + (int64_t)c * lds];
    else v = (r == c && !zero_pad) ? 1.0 : 0.0;
    dst[e] = v;
  }
}

__global__ void copy_unpad_kernel(int N, int Np, const double* __restrict__ src,
                                  double* __restrict__ dst, int64_t ldd) {
  const int64_t total = (int64_t)N * N;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e % N), c = (int)(e / N);
    dst[r + (int64_t)c * ldd] = src[r + (int64_t)c * Np];
  }
}

// Rows/cols >= N of an Np x Np matrix <- identity (after a scatter into a zeroed buffer).
__global__ void set_identity_pad_kernel(int N, int Np, double* __restrict__ A) {
  for (int r = N + blockIdx.x * blockDim.x + threadIdx.x; r < Np; r += gridDim.x * blockDim.x)
    A[(int64_t)r * Np + r] = 1.0;
}

// scal: [0] logdet, [1] ssq, [2] cor, [3] s1, [4] s2  ->  out3 = (f, g1, g2)
__global__ void theta0_finish_kernel(int N, int Np, const double* __restrict__ scal,
                                     double* __restrict__ out3, int want_grad) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    // scal[2] was summed over the padded Np x Np arrays: each pad diagonal entry adds 1 * 1
    out3[0] = 0.5 * ((double)N * 1.8378770664093454835606594728112 + scal[0] + scal[1]) +
              0.5 * (scal[2] - (double)(Np - N));
    out3[1] = want_grad ? -0.5 * scal[3] : 0.0;
    out3[2] = want_grad ? -0.5 * scal[4] : 0.0;
  }
}

static int grid_for(int64_t n);
// ------------------------------------------------------------------ driver helpers
// (sum f, sum g0, sum g1, sum g2) over M individuals by one workgroup in a fixed order: the
// common-HP objective (CF:203-234, 334-368) and the individual part of logL_monitoring.
__global__ __launch_bounds__(256) void sum4_kernel(int64_t M, const double* __restrict__ f,
                                                   const double* __restrict__ g,
                                                   double* __restrict__ out4) {
  __shared__ double red[4][4];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t e = threadIdx.x; e < M; e += 256) {
    v[0] += f[e];
    if (g != nullptr) {
      v[1] += g[3 * e];
      v[2] += g[3 * e + 1];
      v[3] += g[3 * e + 2];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = wave_sum(v[q]);
    if (lane == 0) red[q][wave] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    out4[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
  }
}

__global__ __launch_bounds__(256) void pack_lower_kernel(int Np, double* __restrict__ P,
                                                         double* __restrict__ pack, int unpack) {
  const int p = blockIdx.x;
  int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
  while (i * (i + 1) / 2 > p) --i;
  while ((i + 1) * (i + 2) / 2 <= p) ++i;
  const int j = p - i * (i + 1) / 2;
  double* tile = P + (int64_t)i * NB + (int64_t)j * NB * Np;
  double* buf = pack + (int64_t)p * 4096;
  for (int e = threadIdx.x; e < 4096; e += 256) {
    const int m = e & 63, n = e >> 6;
    if (unpack) tile[m + (int64_t)n * Np] = buf[e];
    else buf[e] = tile[m + (int64_t)n * Np];
  }
}

// tr(A) over the true N x N part (one workgroup, fixed order)
__global__ __launch_bounds__(256) void trace_kernel(int N, int64_t ld, const double* __restrict__ A,
                                                    double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int e = threadIdx.x; e < N; e += 256) s += A[(int64_t)e * (ld + 1)];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

// One individual's view of the hyper-posterior for the dense evaluation route (n_i > 64):
// C (np x np, identity padded) = cov[idx, idx], m (np, zero padded) = mean[idx].
__global__ __launch_bounds__(256) void gather_block_kernel(int n, int np, const int* __restrict__ idx,
                                                           const double* __restrict__ cov, int64_t ldc,
                                                           const double* __restrict__ mean,
                                                           double* __restrict__ C,
                                                           double* __restrict__ m) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = blockIdx.x * 256LL + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e % np), c = (int)(e / np);
    C[e] = (r < n && c < n) ? cov[(int64_t)idx[c] * ldc + idx[r]] : ((r == c) ? 1.0 : 0.0);
  }
  if (blockIdx.x == 0)
    for (int e = threadIdx.x; e < np; e += 256) m[e] = e < n ? mean[idx[e]] : 0.0;
}

int launch_trace(int N, int64_t ld, const double* A, double* out, hipStream_t st) {
  hipLaunchKernelGGL(trace_kernel, dim3(1), dim3(256), 0, st, N, ld, A, out);
  return check_hip(hipGetLastError(), "trace");
}

int launch_gather_block(int n, int np, const int* idx, const double* cov, int64_t ldc,
                        const double* mean, double* C, double* m, hipStream_t st) {
  hipLaunchKernelGGL(gather_block_kernel, dim3(grid_for((int64_t)np * np)), dim3(256), 0, st, n, np, idx,
                     cov, ldc, mean, C, m);
  return check_hip(hipGetLastError(), "gather_block");
}

int launch_sum4(int64_t M, const double* f, const double* g, double* out4, hipStream_t st) {
  hipLaunchKernelGGL(sum4_kernel, dim3(1), dim3(256), 0, st, M, f, g, out4);
  return check_hip(hipGetLastError(), "sum4");
}

int launch_pack_lower(int Np, double* P, double* pack, int unpack, hipStream_t st) {
  const int nt = Np / NB;
  hipLaunchKernelGGL(pack_lower_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, Np, P, pack, unpack);
  return check_hip(hipGetLastError(), "pack_lower");
}

// ------------------------------------------------------------------ launchers
int launch_se_cov_grid(int N, int Np, const double* grid, double th1, double th2, double sigma,
                       double* A, hipStream_t st) {
  dim3 g(Np / 64, Np / 64);
  hipLaunchKernelGGL(se_cov_grid_kernel, g, dim3(256), 0, st, N, Np, grid, th1, th2, sigma, A);
  return check_hip(hipGetLastError(), "se_cov_grid");
}

size_t spd_inverse_workspace_doubles(int Np) { return 3 * (size_t)Np * Np; }

// Panels per super-step of the launch-per-step inverse (dag_workgroups = 0, or more block rows
// than dag_max_block_rows); 1 = the plain right-looking form.
static std::atomic<int> g_panel_blocks{4};
void set_panel_blocks(int v) { g_panel_blocks.store(v < 1 ? 1 : (v > 16 ? 16 : v)); }

int launch_spd_inverse(int Np, double* A, double* logdet, int* info, double* work,
                       hipStream_t st) {
  if (spd_inverse_dag_supported(Np)) return launch_spd_inverse_dag(Np, A, logdet, info, work, st);
  const int nt = Np / NB;
  const int64_t ld = Np;
  double* X = work;
  double* U = work + (size_t)Np * Np;
  double* W = work + 2 * (size_t)Np * Np;
  MAGMA_CHECK_HIP(hipMemsetAsync(logdet, 0, sizeof(double), st));
  MAGMA_CHECK_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
  hipLaunchKernelGGL(potrf_diag0_kernel, dim3(1), dim3(256), 0, st, A, ld, X, U, logdet, info);
  // Super-steps of `pw` panels (block columns): inside one, each new panel's block column is
  // brought up to date with the panels before it (left-looking, rank 64 c) and factored; the
  // matrix right of the super-step is then updated once with all of them (rank 64 pw).
  const int pw = g_panel_blocks.load(std::memory_order_relaxed);
  for (int k0 = 0; k0 + 1 < nt; k0 += pw) {
    const int w = std::min(pw, nt - k0);
    for (int c = 0; c < w; ++c) {
      const int col = k0 + c;
      if (col + 1 >= nt) break;
      hipLaunchKernelGGL(trsm_gemm_kernel, dim3(nt - col - 1), dim3(256), 0, st, A, ld, col, X);
      if (c + 1 < w)  // block column col + 1: panels k0 .. col
        hipLaunchKernelGGL(syrk_update_kernel, dim3(nt - col - 1), dim3(256), 0, st, A, ld, k0, c + 1,
                           col + 1, 1, X, U, logdet, info);
    }
    const int j0 = k0 + w;
    if (j0 < nt) {
      const int rem = nt - j0;
      hipLaunchKernelGGL(syrk_update_kernel, dim3(rem * (rem + 1) / 2), dim3(256), 0, st, A, ld, k0, w, j0,
                         0, X, U, logdet, info);
    }
  }
  int depth = 0;
  while ((1 << depth) < nt) ++depth;
  for (int d = depth - 1; d >= 0; --d) {
    const int nodes = 1 << d;
    const int maxsz = (nt + nodes - 1) / nodes + 1;  // upper bound on a node's block count
    const int half = (maxsz + 1) / 2;
    dim3 g(half, half, nodes);
    hipLaunchKernelGGL(trtri_a_kernel, g, dim3(256), 0, st, A, U, W, ld, nt, d);
    hipLaunchKernelGGL(trtri_b_kernel, g, dim3(256), 0, st, W, X, U, ld, nt, d);
  }
  hipLaunchKernelGGL(lauum_kernel, dim3(nt, nt), dim3(256), 0, st, U, A, ld, nt);
  return check_hip(hipGetLastError(), "spd_inverse");
}

int launch_gemm_nt(int Mr, int Nc, int K, double alpha, const double* A, int64_t lda,
                   const double* B, int64_t ldb, double beta, double* C, int64_t ldc,
                   hipStream_t st) {
  if (Mr % 64 || Nc % 64 || K % 16) return fail(2, "gemm_nt: dimensions must be padded (64/64/16)");
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(Mr / 64, Nc / 64), dim3(256), 0, st, K, alpha, A, lda, B,
                     ldb, beta, C, ldc, 0, 1, 0);
  return check_hip(hipGetLastError(), "gemm_nt");
}

// C = A B^T with B lower triangular (its tiles above the diagonal are never read): half the work.
int launch_gemm_nt_tri(int Np, const double* A, const double* B, double* C, hipStream_t st) {
  if (Np % 64) return fail(2, "gemm_nt_tri: dimension must be padded to 64");
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(Np / 64, Np / 64), dim3(256), 0, st, Np, 1.0, A,
                     (int64_t)Np, B, (int64_t)Np, 0.0, C, (int64_t)Np, 0, 1, 1);
  return check_hip(hipGetLastError(), "gemm_nt_tri");
}

// V = U^T for the upper triangular U (tiles on or above the diagonal; V's tiles above the
// diagonal are left alone): V(n, k) = U(k, n), 64 x 64 tiles through LDS, both sides coalesced.
__global__ __launch_bounds__(256) void transpose_upper_kernel(int nt, const double* __restrict__ U,
                                                              double* __restrict__ V, int64_t ld) {
  __shared__ double Ts[64 * 65];
  const int p = blockIdx.x;  // lower tile (n >= k) of V
  int n = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
  while (n * (n + 1) / 2 > p) --n;
  while ((n + 1) * (n + 2) / 2 <= p) ++n;
  const int k = p - n * (n + 1) / 2;
  const double* Ut = U + (int64_t)k * NB + (int64_t)n * NB * ld;
  double* Vt = V + (int64_t)n * NB + (int64_t)k * NB * ld;
  const int r = threadIdx.x & 63, c0 = threadIdx.x >> 6;
  for (int c = c0; c < 64; c += 4) Ts[c * 65 + r] = Ut[r + (int64_t)c * ld];  // Ts[c][r] = U(r, c)
  __syncthreads();
  for (int c = c0; c < 64; c += 4) Vt[r + (int64_t)c * ld] = Ts[r * 65 + c];  // V(r, c) = U(c, r)
}

int launch_transpose_upper(int Np, const double* U, double* V, hipStream_t st) {
  const int nt = Np / NB;
  hipLaunchKernelGGL(transpose_upper_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, nt, U, V,
                     (int64_t)Np);
  return check_hip(hipGetLastError(), "transpose_upper");
}

// The block rows rank, rank + world, ... of C = A B^T only (sharded theta_0 gradient).
int launch_gemm_nt_rows(int Mr, int Nc, int K, const double* A, int64_t lda, const double* B,
                        int64_t ldb, double* C, int64_t ldc, int rank, int world, hipStream_t st) {
  if (Mr % 64 || Nc % 64 || K % 16) return fail(2, "gemm_nt: dimensions must be padded (64/64/16)");
  const int nt = Mr / 64;
  const int rows = rank < nt ? (nt - rank + world - 1) / world : 0;
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(rows, Nc / 64), dim3(256), 0, st, K, 1.0, A, lda, B, ldb, 0.0,
                     C, ldc, rank, world, 0);
  return check_hip(hipGetLastError(), "gemm_nt_rows");
}

int launch_symv(int Np, const double* A, const double* x, double* yout, hipStream_t st) {
  hipLaunchKernelGGL(symv_kernel, dim3((Np + 3) / 4), dim3(256), 0, st, Np, A, x, yout);
  return check_hip(hipGetLastError(), "symv");
}

static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_axpy(int64_t n, double alpha, const double* x, double* yio, hipStream_t st) {
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, alpha, x, yio);
  return check_hip(hipGetLastError(), "axpy");
}

int launch_fill(int64_t n, double v, double* x, hipStream_t st) {
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, st, n, v, x);
  return check_hip(hipGetLastError(), "fill");
}

int launch_poison_lds(double* sink, hipStream_t st) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (int rc = check_hip(hipGetDevice(&dev), "poison_lds")) return rc;
  if (int rc = check_hip(hipGetDeviceProperties(&prop, dev), "poison_lds")) return rc;
  const size_t bytes = prop.sharedMemPerBlock & ~(size_t)15;  // largest workgroup allocation
  if (int rc = check_hip(hipFuncSetAttribute((const void*)poison_lds_kernel,
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)bytes), "poison_lds"))
    return rc;
  hipLaunchKernelGGL(poison_lds_kernel, dim3(prop.multiProcessorCount * 4), dim3(256), bytes, st,
                     (int)(bytes / 8), sink);
  return check_hip(hipGetLastError(), "poison_lds");
}

// partial: >= 1024 doubles of scratch
int launch_dot(int64_t n, const double* a, const double* b, double* partial, double* out,
               hipStream_t st) {
  int g = grid_for(n);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(dot_stage1_kernel, dim3(g), dim3(256), 0, st, n, a, b, partial);
  hipLaunchKernelGGL(sum_stage2_kernel, dim3(1), dim3(256), 0, st, g, partial, out);
  return check_hip(hipGetLastError(), "dot");
}

int theta0_contract_slices(int Np) {
  const int nt = Np / NB;
  // two k-slices while one slice per lower tile would leave the chip under ~4 workgroups per CU
  static const int forced = [] {
    const char* e = getenv("MAGMA_LOWER_SLICES");
    return e ? atoi(e) : 0;
  }();
  if (forced == 1 || (forced == 2 && (Np / 16) % 2 == 0)) return forced;
  return (nt * (nt + 1) / 2 < 1024 && (Np / 16) % 2 == 0) ? 2 : 1;
}

// Z: theta0_contract_slices(Np) * Np^2 doubles of scratch.
int launch_theta0_contract(int N, int Np, const double* T, const double* Kinv,
                           const double* alpha, const double* grid, double th1, double th2,
                           double* Z, double* partials, double* out2, hipStream_t st,
                           const Theta0Shard* sh, const double* Bprod) {
  if (Bprod == nullptr) Bprod = Kinv;  // Z = T Bprod^T on the lower tiles
  const int nt = Np / NB;
  const int ks = theta0_contract_slices(Np);
  const int64_t n2 = (int64_t)Np * Np;
  const int ntiles = sh ? sh->nplist : nt * (nt + 1) / 2;
  if (ntiles > 0)
    hipLaunchKernelGGL(gemm_nt_lower_kernel, dim3(ntiles * ks), dim3(256), 0, st, Np, T, (int64_t)Np,
                       Bprod, (int64_t)Np, Z, (int64_t)Np, n2, ks, sh ? sh->plist : nullptr);
  hipLaunchKernelGGL(theta0_contract_kernel, dim3(nt, nt), dim3(256), 0, st, N, Np, Z, ks, n2, Kinv,
                     alpha, grid, th1, th2, partials, sh ? sh->rank : 0, sh ? sh->world : 1);
  if (out2 != nullptr)  // (nullptr: the caller sums the per-tile pairs itself)
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(256), 0, st, (int64_t)nt * nt, partials,
                       out2);
  return check_hip(hipGetLastError(), "theta0_contract");
}

int theta0_vec_blocks(int Np) { return (Np + 3) / 4; }

// vpart: 2 * theta0_vec_blocks(Np) doubles.
int launch_theta0_vec(int N, int Np, const double* Kinv, const double* cov, const double* mean,
                      const double* m0, double* alpha, double* vpart, hipStream_t st) {
  hipLaunchKernelGGL(theta0_vec_kernel, dim3(theta0_vec_blocks(Np)), dim3(256), 0, st, N, Np, Kinv,
                     cov, mean, m0, alpha, vpart);
  return check_hip(hipGetLastError(), "theta0_vec");
}

int launch_theta0_finish2(int N, int Np, const double* logdet, const int* info,
                          const double* vpart, const double* pairs, int want_grad, double* out4,
                          hipStream_t st) {
  const int nt = Np / NB;
  hipLaunchKernelGGL(theta0_finish2_kernel, dim3(1), dim3(256), 0, st, N, Np, logdet, info,
                     theta0_vec_blocks(Np), vpart, (int64_t)nt * nt, pairs, want_grad, out4);
  return check_hip(hipGetLastError(), "theta0_finish2");
}

int launch_set_identity_pad(int N, int Np, double* A, hipStream_t st) {
  if (Np > N) hipLaunchKernelGGL(set_identity_pad_kernel, dim3(1), dim3(64), 0, st, N, Np, A);
  return check_hip(hipGetLastError(), "set_identity_pad");
}

int launch_copy_pad(int N, int Np, const double* src, int64_t lds, double* dst, int zero_pad,
                    hipStream_t st) {
  hipLaunchKernelGGL(copy_pad_kernel, dim3(grid_for((int64_t)Np * Np)), dim3(256), 0, st, N, Np,
                     src, lds, dst, zero_pad);
  return check_hip(hipGetLastError(), "copy_pad");
}

int launch_copy_unpad(int N, int Np, const double* src, double* dst, int64_t ldd,
                      hipStream_t st) {
  hipLaunchKernelGGL(copy_unpad_kernel, dim3(grid_for((int64_t)N * N)), dim3(256), 0, st, N, Np,
                     src, dst, ldd);
  return check_hip(hipGetLastError(), "copy_unpad");
}

int launch_theta0_finish(int N, int Np, const double* scal, double* out3, int want_grad,
                         hipStream_t st) {
  hipLaunchKernelGGL(theta0_finish_kernel, dim3(1), dim3(64), 0, st, N, Np, scal, out3,
                     want_grad);
  return check_hip(hipGetLastError(), "theta0_finish");
}

}  // namespace magma
