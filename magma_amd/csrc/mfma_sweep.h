// In-register inverse + log-det of a small SPD matrix held in MFMA accumulator layout, by one
// wavefront: the blocked symmetric sweep shared by the per-individual kernels (small.hip) and
// the pivot blocks of the dense block sweep (dense.hip).
//
// Layout: 16 x 16 tiles, register v of lane (l, g) [l = lane & 15, g = lane >> 4] of tile
// (it, jt) = element (16 it + g + 4 v, 16 jt + l) -- what v_mfma_f64_16x16x4_f64 produces.
// MFMA operands: A fragment lane (l, g) = A[l][g], B fragment lane (l, g) = B[g][l]. Rows
// 4 kk .. 4 kk + 3 of a matrix in accumulator layout ARE its B fragment for k-block kk
// (register kk & 3 of the tiles in tile row kk >> 2).
//
// Sweep (Gauss-Jordan without pivoting on an SPD matrix; pivots = squares of the Cholesky
// diagonal, so log-det and the SPD test are those of a Cholesky). Block J = 4 pivots: with
// P = A[J,J], D = P^-1 (4 x 4, every lane redundantly, LDL'),  A <- A - F' B' - 2 I_J  where
//   B' = A[J,:] with (P - I) in place of P   (registers, by the identity above),
//   F' = B'^T D                               (B' through 2 KB of LDS to transpose it),
// leaves A[i,J] = A[i,J] D, A[J,J] = -D and the Schur complement elsewhere: 16 MFMAs and 12
// LDS instructions per block at NT = 4. After NK blocks the leading 4 NK rows/columns hold
// -A^-1.
// Precondition: diagonal <= 1 (callers scale). The pivot rows/columns come out of the MFMA as
// A[i,J] - F (P - I) and P - (I - D)(P - I) - 2 I, whose rounding residue is eps |A[i,J]| and
// 2 eps in absolute terms: negligible next to F = A[i,J] / P and D = 1 / P only while P <= 1.
#pragma once
#include "common.h"

namespace magma {

typedef double v4d __attribute__((ext_vector_type(4)));

// 1/d by v_rcp_f64 + two Newton steps (<= 2 ulp): keeps a full IEEE division off the
// serial pivot chain.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}

// The callers scale the matrix by a power of two to a largest diagonal entry in [0.5, 1). R's
// solve() (LU with partial pivoting, refused when rcond < eps; CF:133 then takes ginv) is what the
// reference applies. An unpivoted sweep cannot tell a singular matrix by the sign of its pivots:
// the late pivots of a matrix with singular values down to 1e-16 come out as POSITIVE rounding
// noise of ~1e-10 of the scale (measured: sigma = 1e-9, n = 7, length scale e^3; LAPACK's dpotrf
// does the same), and the inverse built from them is finite garbage of norm ~1e10. Two cheap
// tests therefore send every matrix whose COMPUTED condition number is beyond ~1e9 to the slow
// path, which applies R's own rule (LU, rcond, ginv): a pivot (= squared Cholesky diagonal =
// Schur complement) below 2^-30 of the scale, or -- looked at by the callers after the sweep -- a
// diagonal entry of the inverse at or above 2^30. Flagging a well-posed matrix costs time (the
// slow path's LU answers it), never correctness.
constexpr double SWEEP_PIVOT_MIN = 9.313225746154785e-10;  // 2^-30
constexpr double SWEEP_INVDIAG_MAX = 1073741824.0;         // 2^30

// A: NT x NT tiles (identity beyond the matrix). Pn: 64 NT doubles of LDS owned by this
// wavefront. pm, pe accumulate det = pm * 2^pe over the pivots; info = index + 1 of the first
// non-positive pivot (kept if already set). Single wavefront: LDS is in order per wavefront and
// the stores/loads below may alias, so no barrier is needed.
//
// Software pipeline: the serial part of a block (row block -> LDS, 4 x 4 pivot block back, its
// LDL' inverse, this lane's F' fragment: ~35 dependent FP64 operations and two LDS round trips)
// used to sit between the rank-4 updates of consecutive blocks with the matrix core idle. Now
// block b first updates the tile row that holds block b + 1's rows (NT MFMAs), and block b + 1's
// serial part is issued in 12 stages with one of block b's remaining (NT - 1) NT MFMAs after each
// (pinned with sched_barrier: the wavefront issues in order, so a VALU stage runs while the
// MFMA issued before it occupies the matrix core). Same operations on the same operands as the
// unpipelined form: results are bit-identical.
template <int NT, int NK>
__device__ __forceinline__ void mfma_sweep(v4d (&A)[NT][NT], double* Pn, int l, int g,
                                           double& pm, int& pe, int& info) {
  double gm[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    gm[h] = (g == h) ? 1.0 : 0.0;
    asm volatile("" : "+v"(gm[h]));  // opaque: keeps the weights from being folded back to selects
  }
  // serial part of block b: bq = B' fragment (row block with P - I), nf = -F' fragment.
  // hook(k), k = 0 .. 11, is called between its stages.
  auto serial = [&](int b, double (&bq)[NT], double (&nf)[NT], auto&& hook) {
    const int it0 = b >> 2, v0 = b & 3;
    const bool dg = (l == 4 * v0 + g);  // this lane holds a diagonal element of P (tile it0,it0)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      bq[jt] = A[it0][jt][v0] - ((jt == it0 && dg) ? 1.0 : 0.0);
      Pn[(16 * jt + l) * 4 + g] = bq[jt];
    }
    __builtin_amdgcn_wave_barrier();
    hook(0);
    double p[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4 + 2]);
      p[r][0] = lo.x; p[r][1] = lo.y; p[r][2] = hi.x; p[r][3] = hi.y;
      p[r][r] += 1.0;
    }
    hook(1);
    // D = P^-1 (4 x 4 SPD, lower triangle read) by LDL'; bad = index + 1 of the first pivot that is
    // not a positive finite number (what follows a bad pivot is garbage the caller discards: info)
    int bad = 0;
    auto fix = [&](double d, int k) {
      const bool ok = (d > SWEEP_PIVOT_MIN) && (d < 1.7e308);
      bad = (!ok && bad == 0) ? k + 1 : bad;
      return d;
    };
    const double d0 = fix(p[0][0], 0);
    const double r0 = fast_rcp(d0);
    hook(2);
    const double l10 = p[1][0] * r0, l20 = p[2][0] * r0, l30 = p[3][0] * r0;
    const double p11 = fma(-l10, p[1][0], p[1][1]);
    const double p21 = fma(-l20, p[1][0], p[2][1]);
    const double p31 = fma(-l30, p[1][0], p[3][1]);
    double p22 = fma(-l20, p[2][0], p[2][2]);
    double p32 = fma(-l30, p[2][0], p[3][2]);
    double p33 = fma(-l30, p[3][0], p[3][3]);
    hook(3);
    const double d1 = fix(p11, 1);
    const double r1 = fast_rcp(d1);
    hook(4);
    const double l21 = p21 * r1, l31 = p31 * r1;
    p22 = fma(-l21, p21, p22);
    p32 = fma(-l31, p21, p32);
    p33 = fma(-l31, p31, p33);
    const double d2 = fix(p22, 2);
    hook(5);
    const double r2 = fast_rcp(d2);
    const double l32 = p32 * r2;
    p33 = fma(-l32, p32, p33);
    const double d3 = fix(p33, 3);
    hook(6);
    const double r3 = fast_rcp(d3);
    // M = L^-1 (unit lower)
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -fma(l21, m10, l20);
    const double m31 = -fma(l32, m21, l31);
    const double m30 = -fma(l32, m20, fma(l31, m10, l30));
    hook(7);
    // D = M' diag(r) M
    double D[4][4];
    D[3][3] = r3;
    D[3][2] = r3 * m32;
    D[3][1] = r3 * m31;
    D[3][0] = r3 * m30;
    D[2][2] = fma(m32, D[3][2], r2);
    D[2][1] = fma(m32, D[3][1], r2 * m21);
    D[2][0] = fma(m32, D[3][0], r2 * m20);
    D[1][1] = fma(m31, D[3][1], fma(m21, r2 * m21, r1));
    D[1][0] = fma(m31, D[3][0], fma(m21, r2 * m20, r1 * m10));
    D[0][0] = fma(m30, D[3][0], fma(m20, r2 * m20, fma(m10, r1 * m10, r0)));
    D[0][1] = D[1][0]; D[0][2] = D[2][0]; D[0][3] = D[3][0];
    D[1][2] = D[2][1]; D[1][3] = D[3][1]; D[2][3] = D[3][2];
    hook(8);
    info = (bad != 0 && info == 0) ? 4 * b + bad : info;
    {
      int e;
      pm *= frexp(d0, &e); pe += e;
      pm *= frexp(d1, &e); pe += e;
      pm *= frexp(d2, &e); pe += e;
      pm *= frexp(d3, &e); pe += e;
      if ((b & 7) == 7) {  // keep the mantissa product in range
        pm = frexp(pm, &e);
        pe += e;
      }
    }
    // evaluate now: otherwise all pivots stay in registers until the end of the sweep
    asm volatile("" : "+v"(pm), "+v"(pe), "+v"(info));
    hook(9);
    // this lane's column g of D, picked with 0/1 weights (exact). A select chain here is turned
    // into divergent branches around the whole D computation by hipcc.
    double dc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      dc[h] = fma(gm[3], D[h][3], fma(gm[2], D[h][2], fma(gm[1], D[h][1], gm[0] * D[h][0])));
    hook(10);
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4 + 2]);
      nf[it] = -fma(hi.y, dc[3], fma(hi.x, dc[2], fma(lo.y, dc[1], lo.x * dc[0])));
    }
    __builtin_amdgcn_wave_barrier();  // Pn is rewritten by the next block
    hook(11);
  };
  double bq[NT], nf[NT];
  serial(0, bq, nf, [](int) {});
#pragma unroll
  for (int b = 0; b < NK; ++b) {
    const int it0 = b >> 2, v0 = b & 3;
    const bool dg = (l == 4 * v0 + g);
    const int it1 = (b + 1 < NK) ? ((b + 1) >> 2) : it0;  // tile row of the next block's rows
    // phase A: that tile row first
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      A[it1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nf[it1], bq[jt], A[it1][jt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // phase B: the other tile rows, one MFMA per stage of the next block's serial part
    constexpr int CNT = (NT - 1) * NT;
    auto issue = [&](int k) {  // pending MFMAs [k CNT / 12, (k + 1) CNT / 12)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < CNT; ++q) {
        if (q < k * CNT / 12 || q >= (k + 1) * CNT / 12) continue;
        const int ri = q / NT, jt = q % NT;
        const int it = ri < it1 ? ri : ri + 1;
        A[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nf[it], bq[jt], A[it][jt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (b + 1 < NK) {
      double bq2[NT], nf2[NT];
      serial(b + 1, bq2, nf2, issue);
      A[it0][it0][v0] -= dg ? 2.0 : 0.0;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        bq[jt] = bq2[jt];
        nf[jt] = nf2[jt];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) issue(k);
      A[it0][it0][v0] -= dg ? 2.0 : 0.0;
    }
  }
}

// The same sweep on the tiles on and below the diagonal only (A symmetric; A[it][jt], jt <= it: 10
// of 16 tiles and 10 of 16 MFMAs per block at NT = 4, 80 instead of 128 accumulator registers).
// The row block B' = A[J,:] is needed in full: its part left of / on the diagonal tile is in
// registers as before, its part right of it is column J of the tiles BELOW the diagonal tile,
// which the 16 lanes holding those columns write to Pn transposed (Pn's layout [column][h] is what
// the pivot block and the F' fragments are read from anyway) and every lane reads back as its B
// fragment. The update is the same MFMA on the same operands for every tile that is kept, so the
// result is that of the full sweep up to the rounding of the tiles above the diagonal, which the
// full sweep computes separately and this one defines by symmetry.
// Registers of rows beyond 4 NK are zero on entry and stay zero, like the columns beyond it.
template <int NT, int NK>
__device__ __forceinline__ void mfma_sweep_lower(v4d (&A)[NT][NT], double* Pn, int l, int g,
                                                 double& pm, int& pe, int& info) {
  double gm[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    gm[h] = (g == h) ? 1.0 : 0.0;
    asm volatile("" : "+v"(gm[h]));
  }
  auto serial = [&](int b, double (&bq)[NT], double (&nf)[NT], auto&& hook) {
    const int it0 = b >> 2, v0 = b & 3;
    const bool dg = (l == 4 * v0 + g);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (jt > it0) continue;
      bq[jt] = A[it0][jt][v0] - ((jt == it0 && dg) ? 1.0 : 0.0);
      Pn[(16 * jt + l) * 4 + g] = bq[jt];
    }
    if ((l >> 2) == v0) {  // this lane holds one of the columns J of tile column it0
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if (jt <= it0) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          Pn[(16 * jt + g + 4 * v) * 4 + (l & 3)] = A[jt][it0][v];  // (rows >= NR: zeros, written too)
      }
    }
    __builtin_amdgcn_wave_barrier();
    hook(0);
    double p[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4 + 2]);
      p[r][0] = lo.x; p[r][1] = lo.y; p[r][2] = hi.x; p[r][3] = hi.y;
      p[r][r] += 1.0;
    }
    hook(1);
    int bad = 0;
    auto fix = [&](double d, int k) {
      const bool ok = (d > SWEEP_PIVOT_MIN) && (d < 1.7e308);
      bad = (!ok && bad == 0) ? k + 1 : bad;
      return d;
    };
    const double d0 = fix(p[0][0], 0);
    const double r0 = fast_rcp(d0);
    hook(2);
    const double l10 = p[1][0] * r0, l20 = p[2][0] * r0, l30 = p[3][0] * r0;
    const double p11 = fma(-l10, p[1][0], p[1][1]);
    const double p21 = fma(-l20, p[1][0], p[2][1]);
    const double p31 = fma(-l30, p[1][0], p[3][1]);
    double p22 = fma(-l20, p[2][0], p[2][2]);
    double p32 = fma(-l30, p[2][0], p[3][2]);
    double p33 = fma(-l30, p[3][0], p[3][3]);
    hook(3);
    const double d1 = fix(p11, 1);
    const double r1 = fast_rcp(d1);
    hook(4);
    const double l21 = p21 * r1, l31 = p31 * r1;
    p22 = fma(-l21, p21, p22);
    p32 = fma(-l31, p21, p32);
    p33 = fma(-l31, p31, p33);
    const double d2 = fix(p22, 2);
    hook(5);
    const double r2 = fast_rcp(d2);
    const double l32 = p32 * r2;
    p33 = fma(-l32, p32, p33);
    const double d3 = fix(p33, 3);
    hook(6);
    const double r3 = fast_rcp(d3);
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -fma(l21, m10, l20);
    const double m31 = -fma(l32, m21, l31);
    const double m30 = -fma(l32, m20, fma(l31, m10, l30));
    hook(7);
    double D[4][4];
    D[3][3] = r3;
    D[3][2] = r3 * m32;
    D[3][1] = r3 * m31;
    D[3][0] = r3 * m30;
    D[2][2] = fma(m32, D[3][2], r2);
    D[2][1] = fma(m32, D[3][1], r2 * m21);
    D[2][0] = fma(m32, D[3][0], r2 * m20);
    D[1][1] = fma(m31, D[3][1], fma(m21, r2 * m21, r1));
    D[1][0] = fma(m31, D[3][0], fma(m21, r2 * m20, r1 * m10));
    D[0][0] = fma(m30, D[3][0], fma(m20, r2 * m20, fma(m10, r1 * m10, r0)));
    D[0][1] = D[1][0]; D[0][2] = D[2][0]; D[0][3] = D[3][0];
    D[1][2] = D[2][1]; D[1][3] = D[3][1]; D[2][3] = D[3][2];
    hook(8);
    info = (bad != 0 && info == 0) ? 4 * b + bad : info;
    {
      int e;
      pm *= frexp(d0, &e); pe += e;
      pm *= frexp(d1, &e); pe += e;
      pm *= frexp(d2, &e); pe += e;
      pm *= frexp(d3, &e); pe += e;
      if ((b & 7) == 7) {
        pm = frexp(pm, &e);
        pe += e;
      }
    }
    asm volatile("" : "+v"(pm), "+v"(pe), "+v"(info));
    hook(9);
    double dc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      dc[h] = fma(gm[3], D[h][3], fma(gm[2], D[h][2], fma(gm[1], D[h][1], gm[0] * D[h][0])));
    hook(10);
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4 + 2]);
      nf[it] = -fma(hi.y, dc[3], fma(hi.x, dc[2], fma(lo.y, dc[1], lo.x * dc[0])));
      if (it > it0) bq[it] = Pn[(16 * it + l) * 4 + g];  // B' right of the diagonal tile
    }
    __builtin_amdgcn_wave_barrier();  // Pn is rewritten by the next block
    hook(11);
  };
  double bq[NT], nf[NT];
  serial(0, bq, nf, [](int) {});
#pragma unroll
  for (int b = 0; b < NK; ++b) {
    const int it0 = b >> 2, v0 = b & 3;
    const bool dg = (l == 4 * v0 + g);
    const int it1 = (b + 1 < NK) ? ((b + 1) >> 2) : it0;
    // phase A: the tiles the next block's row block comes from -- tile row it1 left of / on the
    // diagonal, tile column it1 below it
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int it = q < it1 ? it1 : q, jt = q < it1 ? q : it1;
      A[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nf[it], bq[jt], A[it][jt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // phase B: the other tiles, spread over the stages of the next block's serial part
    constexpr int CNT = NT * (NT - 1) / 2;
    auto issue = [&](int k) {
      __builtin_amdgcn_sched_barrier(0);
      int q = 0;
#pragma unroll
      for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int jt = 0; jt <= it; ++jt) {
          if (it == it1 || jt == it1) continue;
          if (q >= k * CNT / 12 && q < (k + 1) * CNT / 12)
            A[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nf[it], bq[jt], A[it][jt], 0, 0, 0);
          ++q;
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (b + 1 < NK) {
      double bq2[NT], nf2[NT];
      serial(b + 1, bq2, nf2, issue);
      A[it0][it0][v0] -= dg ? 2.0 : 0.0;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        bq[jt] = bq2[jt];
        nf[jt] = nf2[jt];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 12; ++k) issue(k);
      A[it0][it0][v0] -= dg ? 2.0 : 0.0;
    }
  }
}

}  // namespace magma

namespace magma {

// ------------------------------------------------------------------ blocked Cholesky + L^-1
// M = L4^-1 (lower triangular, 4 x 4) of the Cholesky factor of the SPD block p (lower triangle
// read), and the four pivots d_k = L_kk^2 (the same pivots a column-by-column Cholesky sees:
// log-det and SPD test). LDL' first, then the unit factor's inverse scaled by 1/sqrt(d).
__device__ __forceinline__ double fast_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * fma(-0.5 * d * y, y, 1.5);
  y = y * fma(-0.5 * d * y, y, 1.5);
  return y;
}

__device__ __forceinline__ void chol4_inverse(const double (&p)[4][4], double (&M)[4][4],
                                              double (&piv)[4], int& bad) {
  bad = 0;
  auto chk = [&](double d, int k) {
    const bool ok = (d > 0.0) && (d < 1.7e308);
    bad = (!ok && bad == 0) ? k + 1 : bad;
  };
  const double d0 = p[0][0];
  chk(d0, 0);
  const double r0 = fast_rcp(d0);
  const double l10 = p[1][0] * r0, l20 = p[2][0] * r0, l30 = p[3][0] * r0;
  const double d1 = fma(-l10, p[1][0], p[1][1]);
  const double p21 = fma(-l20, p[1][0], p[2][1]);
  const double p31 = fma(-l30, p[1][0], p[3][1]);
  double p22 = fma(-l20, p[2][0], p[2][2]);
  double p32 = fma(-l30, p[2][0], p[3][2]);
  double p33 = fma(-l30, p[3][0], p[3][3]);
  chk(d1, 1);
  const double r1 = fast_rcp(d1);
  const double l21 = p21 * r1, l31 = p31 * r1;
  p22 = fma(-l21, p21, p22);
  p32 = fma(-l31, p21, p32);
  p33 = fma(-l31, p31, p33);
  const double d2 = p22;
  chk(d2, 2);
  const double r2 = fast_rcp(d2);
  const double l32 = p32 * r2;
  const double d3 = fma(-l32, p32, p33);
  chk(d3, 3);
  piv[0] = d0; piv[1] = d1; piv[2] = d2; piv[3] = d3;
  const double m10 = -l10, m21 = -l21, m32 = -l32;
  const double m20 = -fma(l21, m10, l20);
  const double m31 = -fma(l32, m21, l31);
  const double m30 = -fma(l32, m20, fma(l31, m10, l30));
  const double s0 = fast_rsqrt(d0), s1 = fast_rsqrt(d1), s2 = fast_rsqrt(d2), s3 = fast_rsqrt(d3);
  M[0][0] = s0;       M[0][1] = 0.0;      M[0][2] = 0.0;      M[0][3] = 0.0;
  M[1][0] = s1 * m10; M[1][1] = s1;       M[1][2] = 0.0;      M[1][3] = 0.0;
  M[2][0] = s2 * m20; M[2][1] = s2 * m21; M[2][2] = s2;       M[2][3] = 0.0;
  M[3][0] = s3 * m30; M[3][1] = s3 * m31; M[3][2] = s3 * m32; M[3][3] = s3;
}

// Two wavefronts of one workgroup: right-looking factorisation of the 64 x 64 SPD matrix A
// (accumulator layout; the tiles on and above the diagonal are used) in blocks of 4 columns,
// applied at the same time to B = I, which ends as X = L^-1 (L = the Cholesky factor; lower
// triangular, exact zeros above the diagonal).
// The FACTOR wavefront's per-block chain (pivot block -> its factor -> panel -> the MFMAs that
// produce the next pivot block) is what a block step of the N x N factorisation waits for, so it
// carries only what the chain needs: the factorisation runs in LDL' form (no square roots), and
// everything that can wait -- the 1/sqrt(d) row scaling of X, log-det, the SPD test -- is done by
// the SOLVE wavefront from the pivots FACTOR publishes.
// Block J (rows/columns 4b..4b+3), P = A[J,J] = Lu D Lu' (Lu unit lower), Mu = Lu^-1:
//   FACTOR (owns A):
//     Q = A[:,J] Mu' (rows i >= 4b; A's rows J are already in A-fragment layout by symmetry, the
//         4 x 4 product needs the other lane groups' values: through LDS), Qs = Q D^-1;
//     A -= Qs Q' (= Lp Lp' with the Cholesky panel Lp = Q D^-1/2) on the tiles at or right of J:
//         the tile row that holds the next block's rows first, the rest spread over the next
//         block's serial part;
//     Qs, this lane's row of Mu and the pivots published to SOLVE (slot b & 1 of Xch), barrier.
//   SOLVE (owns B), one block behind:
//     Y[J,:] = Mu B[J,:] (same kind of LDS exchange), B -= Qs Y[J,:] on the rows below,
//     X[J,:] = D^-1/2 Y[J,:] stored over B[J,:]; det *= prod(D), SPD test on D.
// 120 + 80 MFMAs in all; the two chains overlap, so the workgroup's time is the FACTOR chain's.
// Every wavefront of the workgroup must execute the 16 barriers (chol64_idle for the others).
// Pn, Bn: 256 doubles of LDS each (private to FACTOR / SOLVE); Xch: CHOL64_XCH doubles.
constexpr int CHOL64_SLOT = 8 * 64 + 8;     // Qs (4 x 64), Mu rows (4 x 64), 4 pivots
constexpr int CHOL64_XCH = 2 * CHOL64_SLOT;

// How FACTOR hands a block to SOLVE.
// Chol64Barrier: one workgroup barrier per block (every wavefront of the workgroup takes part:
// chol64_idle for those with nothing to do).
struct Chol64Barrier {
  __device__ __forceinline__ bool before_write(int) { return true; }
  __device__ __forceinline__ void published(int) { __syncthreads(); }
  __device__ __forceinline__ bool wait_published(int) {
    __syncthreads();
    return true;
  }
  __device__ __forceinline__ void consumed(int) {}
};

// Wait until the LDS word *flag is >= want (all lanes of the calling wavefront). false: *abort was
// raised by somebody, or this wait gave up (and raised it).
__device__ __forceinline__ bool lds_wait_ge(const int* flag, int want, int* abort_) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= want) break;
    if ((spins & 255u) == 255u) {
      if (spins >= (1u << 22)) __hip_atomic_store(abort_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  return true;
}

__device__ __forceinline__ void lds_publish(int* flag, int value) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Chol64Flags: two counters in LDS (blocks published by FACTOR, blocks taken by SOLVE) and no
// barrier, so that the other wavefronts of the workgroup are free to do something else meanwhile.
// The counters run on from one factorisation to the next (base = 16 x the number done so far), so
// nobody has to reset them; between two factorisations the caller passes a workgroup barrier.
struct Chol64Flags {
  int* pub;
  int* cons;
  int* abort_;
  int base;
  __device__ __forceinline__ bool before_write(int b) {  // slot b & 1 held block b - 2
    return b < 2 || lds_wait_ge(cons, base + b - 1, abort_);
  }
  __device__ __forceinline__ void published(int b) { lds_publish(pub, base + b + 1); }
  __device__ __forceinline__ bool wait_published(int b) { return lds_wait_ge(pub, base + b + 1, abort_); }
  __device__ __forceinline__ void consumed(int b) { lds_publish(cons, base + b + 1); }
};

template <class Sync>
__device__ __forceinline__ bool chol64_factor(v4d (&A)[4][4], double* Pn, double* Xch, int lane,
                                              Sync sy) {
  const int l = lane & 15, g = lane >> 4;
  double gm[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    gm[h] = (g == h) ? 1.0 : 0.0;
    asm volatile("" : "+v"(gm[h]));
  }
  double q[4] = {0.0, 0.0, 0.0, 0.0}, nqs[4] = {0.0, 0.0, 0.0, 0.0};  // block b - 1's Q, -Qs
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int it0 = b >> 2, v0 = b & 3;
    // the previous block's update of the tiles outside this block's tile row: issued a few at a
    // time between the stages below (the matrix core works on them under the pivot chain)
    constexpr int NPEND_MAX = 6;
    int pit[NPEND_MAX], pjt[NPEND_MAX], npend = 0;
    if (b > 0) {
      const int pi0 = (b - 1) >> 2;
#pragma unroll
      for (int it = pi0; it < 4; ++it)
#pragma unroll
        for (int jt = it; jt < 4; ++jt)
          if (it != it0 && (it > pi0 || ((b - 1) & 3) != 3)) {  // (a finished tile row needs no update)
            pit[npend] = it;
            pjt[npend] = jt;
            ++npend;
          }
    }
    auto pending = [&](int k, int nk) {  // share k of nk
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < NPEND_MAX; ++e)
        if (e < npend && e >= k * npend / nk && e < (k + 1) * npend / nk)
          A[pit[e]][pjt[e]] =
              __builtin_amdgcn_mfma_f64_16x16x4f64(nqs[pit[e]], q[pjt[e]], A[pit[e]][pjt[e]], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int jt = it0; jt < 4; ++jt) Pn[(16 * jt + l) * 4 + g] = A[it0][jt][v0];
    __builtin_amdgcn_wave_barrier();
    pending(0, 5);
    double p[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(4 * b + r) * 4 + 2]);
      p[r][0] = lo.x; p[r][1] = lo.y; p[r][2] = hi.x; p[r][3] = hi.y;
    }
    // LDL' of the pivot block (lower triangle read); Mu = Lu^-1
    const double d0 = p[0][0];
    const double r0 = fast_rcp(d0);
    const double l10 = p[1][0] * r0, l20 = p[2][0] * r0, l30 = p[3][0] * r0;
    const double d1 = fma(-l10, p[1][0], p[1][1]);
    const double p21 = fma(-l20, p[1][0], p[2][1]);
    const double p31 = fma(-l30, p[1][0], p[3][1]);
    double p22 = fma(-l20, p[2][0], p[2][2]);
    double p32 = fma(-l30, p[2][0], p[3][2]);
    double p33 = fma(-l30, p[3][0], p[3][3]);
    pending(1, 5);
    const double r1 = fast_rcp(d1);
    const double l21 = p21 * r1, l31 = p31 * r1;
    p22 = fma(-l21, p21, p22);
    p32 = fma(-l31, p21, p32);
    p33 = fma(-l31, p31, p33);
    const double d2 = p22;
    pending(2, 5);
    const double r2 = fast_rcp(d2);
    const double l32 = p32 * r2;
    const double d3 = fma(-l32, p32, p33);
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -fma(l21, m10, l20);
    const double m31 = -fma(l32, m21, l31);
    pending(3, 5);
    const double r3 = fast_rcp(d3);
    const double m30 = -fma(l32, m20, fma(l31, m10, l30));
    // row g of Mu and 1/d_g for this lane (0/1 weights: exact, and no divergent select chains)
    double mr[4];
    mr[0] = fma(gm[3], m30, fma(gm[2], m20, fma(gm[1], m10, gm[0])));
    mr[1] = fma(gm[3], m31, fma(gm[2], m21, gm[1]));
    mr[2] = fma(gm[3], m32, gm[2]);
    mr[3] = gm[3];
    const double rg = fma(gm[3], r3, fma(gm[2], r2, fma(gm[1], r1, gm[0] * r0)));
    const double dg = fma(gm[3], d3, fma(gm[2], d2, fma(gm[1], d1, gm[0] * d0)));
    pending(4, 5);
    double* slot = Xch + CHOL64_SLOT * (b & 1);
    if (!sy.before_write(b)) return false;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (it < it0) {
        q[it] = 0.0;
        nqs[it] = 0.0;
        continue;
      }
      const double2 lo = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Pn[(16 * it + l) * 4 + 2]);
      const double v = fma(hi.y, mr[3], fma(hi.x, mr[2], fma(lo.y, mr[1], lo.x * mr[0])));
      q[it] = (it > it0 || l >= 4 * v0) ? v : 0.0;  // rows above the block are finished
      nqs[it] = -(q[it] * rg);
      slot[it * 64 + lane] = nqs[it];
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) slot[(4 + h) * 64 + lane] = mr[h];
    if (l == 0) slot[512 + g] = dg;
    sy.published(b);  // (barrier b: block b published; the slot of block b-2 may be reused)
    // the tile row of the next block's rows first; the others with the next block (`pending`)
    const int it1 = (b + 1) >> 2;
    if (b + 1 < 16) {
#pragma unroll
      for (int jt = it1; jt < 4; ++jt)
        A[it1][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nqs[it1], q[jt], A[it1][jt], 0, 0, 0);
    }
  }
  return true;
}

__device__ __forceinline__ void chol64_factor(v4d (&A)[4][4], double* Pn, double* Xch, int lane) {
  (void)chol64_factor(A, Pn, Xch, lane, Chol64Barrier());
}

// pm, pe accumulate det = pm * 2^pe over the pivots; info = index + 1 of the first pivot that is
// not a positive finite number (kept if already set).
// row_out(b, jt, x): the finished rows 4b .. 4b+3 of X as block b leaves them (this lane: row
// 4b + g, column 16 jt + l; every jt, zeros above the diagonal), for callers that want X before
// the last block is done.
template <class Sync, class RowOut>
__device__ __forceinline__ bool chol64_solve(v4d (&B)[4][4], double* Bn, const double* Xch,
                                             int lane, double& pm, int& pe, int& info, Sync sy,
                                             RowOut&& row_out) {
  const int l = lane & 15, g = lane >> 4;
  double gm[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    gm[h] = (g == h) ? 1.0 : 0.0;
    asm volatile("" : "+v"(gm[h]));
  }
#pragma unroll
  for (int it = 0; it < 4; ++it)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int v = 0; v < 4; ++v) B[it][jt][v] = (it == jt && g + 4 * v == l) ? 1.0 : 0.0;
  double dsave = 1.0;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int it0 = b >> 2, v0 = b & 3;
#pragma unroll
    for (int jt = 0; jt <= it0; ++jt) Bn[(16 * jt + l) * 4 + g] = B[it0][jt][v0];
    if (!sy.wait_published(b)) return false;  // (barrier b)
    const double* slot = Xch + CHOL64_SLOT * (b & 1);
    double mr[4], nqs[4], yr[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) mr[h] = slot[(4 + h) * 64 + lane];
#pragma unroll
    for (int it = it0; it < 4; ++it) nqs[it] = slot[it * 64 + lane];
    const double2 dlo = *reinterpret_cast<const double2*>(&slot[512]);
    const double2 dhi = *reinterpret_cast<const double2*>(&slot[514]);
    const double d[4] = {dlo.x, dlo.y, dhi.x, dhi.y};
    sy.consumed(b);
#pragma unroll
    for (int jt = 0; jt <= it0; ++jt) {
      const double2 lo = *reinterpret_cast<const double2*>(&Bn[(16 * jt + l) * 4]);
      const double2 hi = *reinterpret_cast<const double2*>(&Bn[(16 * jt + l) * 4 + 2]);
      yr[jt] = fma(hi.y, mr[3], fma(hi.x, mr[2], fma(lo.y, mr[1], lo.x * mr[0])));
    }
    __builtin_amdgcn_wave_barrier();  // Bn is rewritten by the next block
#pragma unroll
    for (int it = it0; it < 4; ++it)
#pragma unroll
      for (int jt = 0; jt <= it0; ++jt)
        B[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(nqs[it], yr[jt], B[it][jt], 0, 0, 0);
    // off the FACTOR chain: the 1/sqrt(d) scaling of the finished rows; the pivots are kept for
    // the SPD test and the determinant, done once at the end with one pivot per lane (block by
    // block they cost this wavefront ~40 serial operations per block, and FACTOR waits for a
    // SOLVE that falls two blocks behind)
    const double dg = fma(gm[3], d[3], fma(gm[2], d[2], fma(gm[1], d[1], gm[0] * d[0])));
    dsave = (l == b) ? dg : dsave;  // lane (l = b, g) keeps pivot 4 b + g
    const double sg = fast_rsqrt(dg);
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      if (jt <= it0) B[it0][jt][v0] = yr[jt] * sg;
      row_out(b, jt, jt <= it0 ? (double)B[it0][jt][v0] : 0.0);
    }
  }
  {
    // lane (l, g) holds pivot p = 4 l + g: first bad pivot by ballot, det = product of mantissas
    // x 2^(sum of exponents) over the wavefront in a fixed order
    const bool ok = (dsave > 0.0) && (dsave < 1.7e308);
    const int p = 4 * l + g;
    int first = ok ? 64 : p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(first, o);
      first = other < first ? other : first;
    }
    info = (first < 64 && info == 0) ? first + 1 : info;
    int e;
    double m = frexp(dsave, &e);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m *= __shfl_xor(m, o);
      e += __shfl_xor(e, o);
    }
    int e2;
    pm *= frexp(m, &e2);
    pe += e + e2;
  }
  return true;
}

__device__ __forceinline__ void chol64_solve(v4d (&B)[4][4], double* Bn, const double* Xch,
                                             int lane, double& pm, int& pe, int& info) {
  (void)chol64_solve(B, Bn, Xch, lane, pm, pe, info, Chol64Barrier(), [](int, int, double) {});
}

__device__ __forceinline__ void chol64_idle() {
#pragma unroll 1
  for (int b = 0; b < 16; ++b) __syncthreads();
}

}  // namespace magma
