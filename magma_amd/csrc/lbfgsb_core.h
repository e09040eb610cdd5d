// L-BFGS-B (unbounded case) for one small problem as a reverse-communication state machine,
// compiled for the host (lbfgsb.cpp: batched lock-step driver) and for the device (small.hip:
// one wavefront optimises its individual without leaving the GPU). One source, so that both
// walk through identical arithmetic (-ffp-contract=off on both sides; sqrt and division are
// correctly rounded on both).
//
// Stands in for `optimr::opm(..., method = "L-BFGS-B", control = list(kkt = FALSE))`, i.e.
// stats::optim's L-BFGS-B with its defaults (Computing_functions.R:412-413, 417-418, 426-429;
// MAGMA.R:93-94). optimr / R's lbfgsb.c are not part of the reference tree; this is a
// restatement of the published algorithm for the case the reference uses (no bounds):
//   * Byrd, Lu, Nocedal, Zhu (1995), "A limited memory algorithm for bound constrained
//     optimization": with no bounds the generalized Cauchy point is x itself once a
//     correction pair exists and the subspace step is the limited-memory quasi-Newton step
//     -H g with H0 = (s'y / y'y) I -- computed here by the two-loop recursion; the first
//     iteration is steepest descent with initial step min(1/||d||, stpmx).
//   * More & Thuente (1994) line search (MINPACK-2 dcsrch/dcstep) with ftol 1e-3, gtol 0.9,
//     xtol 0.1, at most 20 evaluations; on failure the memory is dropped and the iteration
//     restarted from steepest descent; failure with empty memory ends the run.
//   * stop when (f_old - f) <= factr * eps * max(|f_old|, |f|, 1), when the gradient's
//     max-norm <= pgtol, or after maxit iterations (R counts NEW_X events and stops when the
//     count exceeds maxit); a correction pair is skipped when s'y <= eps * (-g_old'd * stp).
//   * R's driver raises an error on a non-finite objective value at the start; here that
//     problem stops with status 5 and its result is NaN, which the EM driver treats like the
//     reference's `anyNA(new_hp)` guard (MAGMA.R:46-51). A non-finite trial value inside a line
//     search bisects the step towards the best point so far (R's LU-based solve() returns
//     finite garbage there and the search backs off).
#pragma once
#include <math.h>

#ifdef __HIP__  // compiled as HIP (small.hip); lbfgsb.cpp is plain C++
#define LB_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LB_HD inline
#endif

namespace lbcore {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double STPMX = 1e10;
constexpr double FTOL = 1e-3, GTOL = 0.9, XTOL = 0.1;
constexpr int MAXLS = 20;

enum Task { TASK_FG, TASK_CONV, TASK_WARN, TASK_ERROR };

LB_HD double dmax(double a, double b) { return a < b ? b : a; }  // std::max(a, b) semantics
LB_HD double dmin(double a, double b) { return b < a ? b : a; }  // std::min(a, b) semantics
LB_HD double dmax3(double a, double b, double c) { return dmax(dmax(a, b), c); }
LB_HD bool finite_d(double v) { return (v - v) == 0.0; }

struct LineSearch {
  bool brackt;
  int stage;
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
  double cap;  // smallest step at which the objective was non-finite (+inf if none)
};

// MINPACK-2 dcstep: safeguarded cubic/quadratic step, updates the interval of uncertainty.
LB_HD void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy,
                  double& stp, double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / fabs(dx));
  double stpf;
  if (fp > fx) {  // case 1: higher function value, minimum bracketed
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = dmax3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {  // case 2: derivatives of opposite sign
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = dmax3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = true;
  } else if (fabs(dp) < fabs(dx)) {  // case 3: derivative magnitude decreases
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = dmax3(fabs(theta), fabs(dx), fabs(dp));
    double gamma = s * sqrt(dmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (dx - dp)) + gamma;
    const double r = p / q;
    double stpc;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = dmin(stp + 0.66 * (sty - stp), stpf);
      else stpf = dmax(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = dmin(stpmax, stpf);
      stpf = dmax(stpmin, stpf);
    }
  } else {  // case 4: derivative magnitude does not decrease
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = dmax3(fabs(theta), fabs(dy), fabs(dp));
      double gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dy;
      const double r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
}

LB_HD void dcsrch_start(LineSearch& ls, double f, double g, double stp) {
  ls.brackt = false;
  ls.stage = 1;
  ls.finit = f;
  ls.ginit = g;
  ls.gtest = FTOL * ls.ginit;
  ls.width = STPMX - 0.0;
  ls.width1 = ls.width / 0.5;
  ls.stx = 0.0;
  ls.fx = ls.finit;
  ls.gx = ls.ginit;
  ls.sty = 0.0;
  ls.fy = ls.finit;
  ls.gy = ls.ginit;
  ls.stmin = 0.0;
  ls.stmax = stp + 4.0 * stp;
  ls.cap = HUGE_VAL;
}

// One More-Thuente iteration with the values (f, g) at step stp; returns the next task and,
// for TASK_FG, the next trial step in stp.
LB_HD Task dcsrch_step(LineSearch& ls, double f, double g, double& stp) {
  const double stpmin = 0.0, stpmax = STPMX;
  const double ftest = ls.finit + stp * ls.gtest;
  if (ls.stage == 1 && f <= ftest && g >= 0.0) ls.stage = 2;
  Task task = TASK_FG;
  if (ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) task = TASK_WARN;  // rounding errors
  if (ls.brackt && ls.stmax - ls.stmin <= XTOL * ls.stmax) task = TASK_WARN;  // xtol
  if (stp == stpmax && f <= ftest && g <= ls.gtest) task = TASK_WARN;
  if (stp == stpmin && (f > ftest || g >= ls.gtest)) task = TASK_WARN;
  if (f <= ftest && fabs(g) <= GTOL * (-ls.ginit)) task = TASK_CONV;
  if (task != TASK_FG) return task;
  if (ls.stage == 1 && f <= ls.fx && f > ftest) {
    const double fm = f - stp * ls.gtest;
    double fxm = ls.fx - ls.stx * ls.gtest;
    double fym = ls.fy - ls.sty * ls.gtest;
    const double gm = g - ls.gtest;
    double gxm = ls.gx - ls.gtest;
    double gym = ls.gy - ls.gtest;
    dcstep(ls.stx, fxm, gxm, ls.sty, fym, gym, stp, fm, gm, ls.brackt, ls.stmin, ls.stmax);
    ls.fx = fxm + ls.stx * ls.gtest;
    ls.fy = fym + ls.sty * ls.gtest;
    ls.gx = gxm + ls.gtest;
    ls.gy = gym + ls.gtest;
  } else {
    dcstep(ls.stx, ls.fx, ls.gx, ls.sty, ls.fy, ls.gy, stp, f, g, ls.brackt, ls.stmin, ls.stmax);
  }
  if (ls.brackt) {
    if (fabs(ls.sty - ls.stx) >= 0.66 * ls.width1) stp = ls.stx + 0.5 * (ls.sty - ls.stx);
    ls.width1 = ls.width;
    ls.width = fabs(ls.sty - ls.stx);
  }
  if (ls.brackt) {
    ls.stmin = dmin(ls.stx, ls.sty);
    ls.stmax = dmax(ls.stx, ls.sty);
  } else {
    ls.stmin = stp + 1.1 * (stp - ls.stx);
    ls.stmax = stp + 4.0 * (stp - ls.stx);
  }
  stp = dmax(stp, stpmin);
  stp = dmin(stp, stpmax);
  if ((ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) ||
      (ls.brackt && ls.stmax - ls.stmin <= XTOL * ls.stmax))
    stp = ls.stx;
  return TASK_FG;
}

// State of one problem with up to DCAP variables and MCAP correction pairs.
template <int DCAP, int MCAP>
struct State {
  int d, m;
  int status;  // 0 running, 1 rel. reduction, 2 pgtol, 3 maxit, 4 line search, 5 non-finite
  int phase;   // 0: awaiting f,g at x0; 1: awaiting f,g at the line-search trial point
  int iter, nfev, col, head, ifun, iback;
  double f, fold, theta, stp, dnorm, gd, gdold;
  LineSearch ls;
  double x[DCAP], g[DCAP], xold[DCAP], gold[DCAP], dir[DCAP], xtrial[DCAP];
  double S[MCAP][DCAP], Y[MCAP][DCAP], rho[MCAP];
};

template <int DCAP, int MCAP>
LB_HD void init(State<DCAP, MCAP>& s, int d, int m, const double* x0) {
  s.d = d;
  s.m = m;
  s.status = 0;
  s.phase = 0;
  s.iter = s.nfev = s.col = s.head = s.ifun = s.iback = 0;
  s.f = s.fold = 0.0;
  s.theta = 1.0;
  s.stp = s.dnorm = s.gd = s.gdold = 0.0;
  for (int i = 0; i < d; ++i) {
    s.x[i] = x0[i];
    s.xtrial[i] = x0[i];
    s.g[i] = s.xold[i] = s.gold[i] = s.dir[i] = 0.0;
  }
}

// DFIX / MFIX: the number of variables / of correction pairs when the caller knows them at
// compile time (0: read from the state). Same operations in the same order either way; with
// constants the loops unroll, the slot arithmetic loses its integer divisions and the loads of the
// state can be issued together -- on the device, where the state lives in LDS and every wavefront
// steps its optimiser between two evaluations, that is 13k -> ~6k cycles per step.
template <int DFIX, int DCAP, int MCAP>
LB_HD double dot(const State<DCAP, MCAP>& s, const double* a, const double* b) {
  const int d = DFIX ? DFIX : s.d;
  double r = 0.0;
  for (int i = 0; i < d; ++i) r += a[i] * b[i];
  return r;
}

template <int DFIX, int DCAP, int MCAP>
LB_HD void set_trial(State<DCAP, MCAP>& s) {
  const int d = DFIX ? DFIX : s.d;
  if (s.stp == 1.0) for (int i = 0; i < d; ++i) s.xtrial[i] = s.xold[i] + s.dir[i];
  else for (int i = 0; i < d; ++i) s.xtrial[i] = s.stp * s.dir[i] + s.xold[i];
}

// Direction + line-search initialisation at the current iterate. Returns false if the run for
// this problem ended (abnormal line search).
template <int DFIX, int MFIX, int DCAP, int MCAP>
LB_HD bool begin_iteration(State<DCAP, MCAP>& s) {
  const int d = DFIX ? DFIX : s.d, m = MFIX ? MFIX : s.m;
  for (;;) {
    if (s.col == 0) {
      for (int i = 0; i < d; ++i) s.dir[i] = -s.g[i];
    } else {
      double q[DCAP], a[MCAP];
      for (int i = 0; i < d; ++i) q[i] = s.g[i];
      // (k over the capacity with a guard, not over col: unrolled, a[] stays in registers)
#pragma unroll
      for (int k = MCAP - 1; k >= 0; --k) {
        if (k >= s.col) continue;
        const int slot = (s.head + k) % m;
        a[k] = s.rho[slot] * dot<DFIX>(s, s.S[slot], q);
        for (int i = 0; i < d; ++i) q[i] -= a[k] * s.Y[slot][i];
      }
      for (int i = 0; i < d; ++i) q[i] /= s.theta;
#pragma unroll
      for (int k = 0; k < MCAP; ++k) {
        if (k >= s.col) continue;
        const int slot = (s.head + k) % m;
        const double beta = s.rho[slot] * dot<DFIX>(s, s.Y[slot], q);
        for (int i = 0; i < d; ++i) q[i] += s.S[slot][i] * (a[k] - beta);
      }
      for (int i = 0; i < d; ++i) s.dir[i] = -q[i];
    }
    const double dtd = dot<DFIX>(s, s.dir, s.dir);
    s.dnorm = sqrt(dtd);
    s.stp = (s.iter == 0) ? dmin(1.0 / s.dnorm, STPMX) : 1.0;
    for (int i = 0; i < d; ++i) {
      s.xold[i] = s.x[i];
      s.gold[i] = s.g[i];
    }
    s.fold = s.f;
    s.ifun = 0;
    s.iback = 0;
    s.gd = dot<DFIX>(s, s.g, s.dir);
    s.gdold = s.gd;
    if (!(s.gd < 0.0)) {  // ascent direction in projection
      if (s.col == 0) {
        s.status = 4;
        return false;
      }
      s.col = 0;
      s.head = 0;
      s.theta = 1.0;
      continue;
    }
    dcsrch_start(s.ls, s.f, s.gd, s.stp);
    s.ifun = 1;
    s.iback = 0;
    set_trial<DFIX>(s);
    s.phase = 1;
    return true;
  }
}

// Feed f, g at the point last asked for (x0 in phase 0, xtrial afterwards). On return either
// status != 0 (finished: result in x, f) or xtrial holds the next point to evaluate.
template <int DFIX, int MFIX, int DCAP, int MCAP>
LB_HD void tell_fixed(State<DCAP, MCAP>& s, double fv, const double* gv, double factr, double pgtol,
                      int maxit) {
  const int d = DFIX ? DFIX : s.d, m = MFIX ? MFIX : s.m;
  s.nfev++;
  bool finite = finite_d(fv);
  for (int i = 0; i < d && finite; ++i) finite = finite_d(gv[i]);
  if (!finite) {
    if (s.phase == 0) {  // cannot start: R's driver raises an error here
      s.status = 5;
      return;
    }
    s.ls.cap = dmin(s.ls.cap, s.stp);
    s.stp = s.ls.stx + 0.5 * (s.stp - s.ls.stx);
    s.ifun++;
    s.iback = s.ifun - 1;
    if (s.iback >= MAXLS) {
      if (s.col == 0) {
        s.status = 4;
        return;
      }
      s.col = 0;
      s.head = 0;
      s.theta = 1.0;
      begin_iteration<DFIX, MFIX>(s);
      return;
    }
    set_trial<DFIX>(s);
    return;
  }
  if (s.phase == 0) {
    s.f = fv;
    double sb = 0.0;
    for (int i = 0; i < d; ++i) {
      s.g[i] = gv[i];
      sb = dmax(sb, fabs(gv[i]));
    }
    if (sb <= pgtol) {
      s.status = 2;
      return;
    }
    begin_iteration<DFIX, MFIX>(s);
    return;
  }
  // line-search evaluation
  s.gd = dot<DFIX>(s, gv, s.dir);
  double stp = s.stp;
  const Task task = dcsrch_step(s.ls, fv, s.gd, stp);
  if (task == TASK_FG) {
    if (stp >= s.ls.cap) stp = s.ls.stx + 0.5 * (s.ls.cap - s.ls.stx);
    s.stp = stp;
    s.ifun++;
    s.iback = s.ifun - 1;
    if (s.iback >= MAXLS) {
      // either restart from steepest descent or give up (x, g, f still hold the iterate the
      // search started from)
      if (s.col == 0) {
        s.status = 4;
        return;
      }
      s.col = 0;
      s.head = 0;
      s.theta = 1.0;
      begin_iteration<DFIX, MFIX>(s);
      return;
    }
    set_trial<DFIX>(s);
    return;
  }
  // accepted (CONVERGENCE or WARNING): trial point becomes the new iterate
  double sb = 0.0;
  for (int i = 0; i < d; ++i) {
    s.x[i] = s.xtrial[i];
    s.g[i] = gv[i];
    sb = dmax(sb, fabs(gv[i]));
  }
  s.f = fv;
  s.iter++;
  if (s.iter > maxit) {
    s.status = 3;
    return;
  }
  if (sb <= pgtol) {
    s.status = 2;
    return;
  }
  const double ddum = dmax3(fabs(s.fold), fabs(s.f), 1.0);
  if ((s.fold - s.f) <= EPSMCH * factr * ddum) {
    s.status = 1;
    return;
  }
  // correction pair: y = g - g_old, s = stp * d
  double yy[DCAP], ss[DCAP];
  for (int i = 0; i < d; ++i) yy[i] = gv[i] - s.gold[i];
  const double rr = dot<DFIX>(s, yy, yy);
  double dr, dd2;
  if (s.stp == 1.0) {
    dr = s.gd - s.gdold;
    dd2 = -s.gdold;
    for (int i = 0; i < d; ++i) ss[i] = s.dir[i];
  } else {
    dr = (s.gd - s.gdold) * s.stp;
    dd2 = -s.gdold * s.stp;
    for (int i = 0; i < d; ++i) ss[i] = s.stp * s.dir[i];
  }
  if (!(dr <= EPSMCH * dd2)) {
    int slot;
    if (s.col < m) {
      slot = (s.head + s.col) % m;
      s.col++;
    } else {
      slot = s.head;
      s.head = (s.head + 1) % m;
    }
    for (int i = 0; i < d; ++i) {
      s.S[slot][i] = ss[i];
      s.Y[slot][i] = yy[i];
    }
    s.rho[slot] = 1.0 / dr;
    s.theta = rr / dr;
  }
  begin_iteration<DFIX, MFIX>(s);
}

template <int DCAP, int MCAP>
LB_HD void tell(State<DCAP, MCAP>& s, double fv, const double* gv, double factr, double pgtol,
                int maxit) {
  tell_fixed<0, 0>(s, fv, gv, factr, pgtol, maxit);
}

}  // namespace lbcore
