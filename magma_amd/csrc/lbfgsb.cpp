// Batched, lock-step L-BFGS-B (unbounded case) with reverse communication -- host code.
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
//   * R's driver raises an error on a non-finite objective value; here that problem stops
//     with status 5 and its result is NaN, which the EM driver treats like the reference's
//     `anyNA(new_hp)` guard (MAGMA.R:46-51).
// B independent problems advance together so that one batched objective evaluation (one GPU
// launch over all individuals) serves every problem's current trial point.
#include "../../include/magma_hip.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double STPMX = 1e10;
constexpr double FTOL = 1e-3, GTOL = 0.9, XTOL = 0.1;
constexpr int MAXLS = 20;

enum Task { TASK_FG, TASK_CONV, TASK_WARN, TASK_ERROR };

struct LineSearch {
  bool brackt;
  int stage;
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
  double cap;  // smallest step at which the objective was non-finite (+inf if none)
};

// MINPACK-2 dcstep: safeguarded cubic/quadratic step, updates the interval of uncertainty.
void dcstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
            double fp, double dp, bool& brackt, double stpmin, double stpmax) {
  const double sgnd = dp * (dx / std::fabs(dx));
  double stpf;
  if (fp > fx) {  // case 1: higher function value, minimum bracketed
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (std::fabs(stpc - stx) < std::fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {  // case 2: derivatives of opposite sign
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (std::fabs(stpc - stp) > std::fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = true;
  } else if (std::fabs(dp) < std::fabs(dx)) {  // case 3: derivative magnitude decreases
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = std::max({std::fabs(theta), std::fabs(dx), std::fabs(dp)});
    double gamma = s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
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
      if (std::fabs(stpc - stp) < std::fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = std::min(stp + 0.66 * (sty - stp), stpf);
      else stpf = std::max(stp + 0.66 * (sty - stp), stpf);
    } else {
      if (std::fabs(stpc - stp) > std::fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = std::min(stpmax, stpf);
      stpf = std::max(stpmin, stpf);
    }
  } else {  // case 4: derivative magnitude does not decrease
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = std::max({std::fabs(theta), std::fabs(dy), std::fabs(dp)});
      double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
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

void dcsrch_start(LineSearch& ls, double f, double g, double stp) {
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
  ls.cap = std::numeric_limits<double>::infinity();
}

// One More-Thuente iteration with the values (f, g) at step stp; returns the next task and,
// for TASK_FG, the next trial step in stp.
Task dcsrch_step(LineSearch& ls, double f, double g, double& stp) {
  const double stpmin = 0.0, stpmax = STPMX;
  const double ftest = ls.finit + stp * ls.gtest;
  if (ls.stage == 1 && f <= ftest && g >= 0.0) ls.stage = 2;
  Task task = TASK_FG;
  if (ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) task = TASK_WARN;  // rounding errors
  if (ls.brackt && ls.stmax - ls.stmin <= XTOL * ls.stmax) task = TASK_WARN;  // xtol
  if (stp == stpmax && f <= ftest && g <= ls.gtest) task = TASK_WARN;
  if (stp == stpmin && (f > ftest || g >= ls.gtest)) task = TASK_WARN;
  if (f <= ftest && std::fabs(g) <= GTOL * (-ls.ginit)) task = TASK_CONV;
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
    if (std::fabs(ls.sty - ls.stx) >= 0.66 * ls.width1) stp = ls.stx + 0.5 * (ls.sty - ls.stx);
    ls.width1 = ls.width;
    ls.width = std::fabs(ls.sty - ls.stx);
  }
  if (ls.brackt) {
    ls.stmin = std::min(ls.stx, ls.sty);
    ls.stmax = std::max(ls.stx, ls.sty);
  } else {
    ls.stmin = stp + 1.1 * (stp - ls.stx);
    ls.stmax = stp + 4.0 * (stp - ls.stx);
  }
  stp = std::max(stp, stpmin);
  stp = std::min(stp, stpmax);
  if ((ls.brackt && (stp <= ls.stmin || stp >= ls.stmax)) ||
      (ls.brackt && ls.stmax - ls.stmin <= XTOL * ls.stmax))
    stp = ls.stx;
  return TASK_FG;
}

struct Problem {
  int status = 0;  // 0 running, 1 rel. reduction, 2 pgtol, 3 maxit, 4 line search, 5 non-finite
  int phase = 0;   // 0: awaiting f,g at x0; 1: awaiting f,g at the line-search trial point
  int iter = 0, nfev = 0, col = 0, head = 0, ifun = 0, iback = 0;
  double f = 0, fold = 0, theta = 1.0, stp = 0, dnorm = 0, gd = 0, gdold = 0;
  LineSearch ls;
};

}  // namespace

struct magma_lbfgsb {
  int B, d, m, maxit;
  double factr, pgtol;
  std::vector<Problem> pr;
  std::vector<double> x, g, xold, gold, dir, xtrial, S, Y, rho;
  std::vector<int> act;  // problems still running, ascending (compact ask/tell)

  double* vx(std::vector<double>& v, int b) { return v.data() + (size_t)b * d; }
  double* sp(int b, int slot) { return S.data() + ((size_t)b * m + slot) * d; }
  double* yp(int b, int slot) { return Y.data() + ((size_t)b * m + slot) * d; }

  double dot(const double* a, const double* b_) const {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += a[i] * b_[i];
    return s;
  }

  void set_trial(int b) {
    Problem& p = pr[b];
    double* xt = vx(xtrial, b);
    const double *xo = vx(xold, b), *dd = vx(dir, b);
    if (p.stp == 1.0) for (int i = 0; i < d; ++i) xt[i] = xo[i] + dd[i];
    else for (int i = 0; i < d; ++i) xt[i] = p.stp * dd[i] + xo[i];
  }

  // Direction + line-search initialisation at the current iterate. Returns false if the run
  // for this problem ended (abnormal line search).
  bool begin_iteration(int b) {
    Problem& p = pr[b];
    for (;;) {
      double* dd = vx(dir, b);
      const double* gg = vx(g, b);
      if (p.col == 0) {
        for (int i = 0; i < d; ++i) dd[i] = -gg[i];
      } else {
        double q[64], a[64];
        std::vector<double> qv;
        double* qq = q;
        if (d > 64) { qv.resize(d); qq = qv.data(); }
        for (int i = 0; i < d; ++i) qq[i] = gg[i];
        for (int k = p.col - 1; k >= 0; --k) {
          const int slot = (p.head + k) % m;
          a[k] = rho[(size_t)b * m + slot] * dot(sp(b, slot), qq);
          const double* y = yp(b, slot);
          for (int i = 0; i < d; ++i) qq[i] -= a[k] * y[i];
        }
        for (int i = 0; i < d; ++i) qq[i] /= p.theta;
        for (int k = 0; k < p.col; ++k) {
          const int slot = (p.head + k) % m;
          const double beta = rho[(size_t)b * m + slot] * dot(yp(b, slot), qq);
          const double* s = sp(b, slot);
          for (int i = 0; i < d; ++i) qq[i] += s[i] * (a[k] - beta);
        }
        for (int i = 0; i < d; ++i) dd[i] = -qq[i];
      }
      const double dtd = dot(dd, dd);
      p.dnorm = std::sqrt(dtd);
      p.stp = (p.iter == 0) ? std::min(1.0 / p.dnorm, STPMX) : 1.0;
      std::copy(vx(x, b), vx(x, b) + d, vx(xold, b));
      std::copy(gg, gg + d, vx(gold, b));
      p.fold = p.f;
      p.ifun = 0;
      p.iback = 0;
      p.gd = dot(gg, dd);
      p.gdold = p.gd;
      if (!(p.gd < 0.0)) {  // ascent direction in projection
        if (p.col == 0) {
          p.status = 4;
          return false;
        }
        p.col = 0;
        p.head = 0;
        p.theta = 1.0;
        continue;
      }
      dcsrch_start(p.ls, p.f, p.gd, p.stp);
      p.ifun = 1;
      p.iback = 0;
      set_trial(b);
      p.phase = 1;
      return true;
    }
  }

  void tell_one(int b, double fv, const double* gv) {
    Problem& p = pr[b];
    p.nfev++;
    bool finite = std::isfinite(fv);
    for (int i = 0; i < d && finite; ++i) finite = std::isfinite(gv[i]);
    if (!finite) {
      if (p.phase == 0) {  // cannot start: R's driver raises an error here
        p.status = 5;
        return;
      }
      // Non-finite trial point inside a line search (a covariance that is not numerically
      // positive definite). R's LU-based solve() returns finite garbage there and the search
      // backs off; here the step is bisected towards the best point so far and capped.
      p.ls.cap = std::min(p.ls.cap, p.stp);
      p.stp = p.ls.stx + 0.5 * (p.stp - p.ls.stx);
      p.ifun++;
      p.iback = p.ifun - 1;
      if (p.iback >= MAXLS) {
        if (p.col == 0) {
          p.status = 4;
          return;
        }
        p.col = 0;
        p.head = 0;
        p.theta = 1.0;
        begin_iteration(b);
        return;
      }
      set_trial(b);
      return;
    }
    if (p.phase == 0) {
      p.f = fv;
      std::copy(gv, gv + d, vx(g, b));
      double sb = 0.0;
      for (int i = 0; i < d; ++i) sb = std::max(sb, std::fabs(gv[i]));
      if (sb <= pgtol) {
        p.status = 2;
        return;
      }
      begin_iteration(b);
      return;
    }
    // line-search evaluation
    const double* dd = vx(dir, b);
    p.gd = dot(gv, dd);
    double stp = p.stp;
    const Task task = dcsrch_step(p.ls, fv, p.gd, stp);
    if (task == TASK_FG) {
      if (stp >= p.ls.cap) stp = p.ls.stx + 0.5 * (p.ls.cap - p.ls.stx);
      p.stp = stp;
      p.ifun++;
      p.iback = p.ifun - 1;
      if (p.iback >= MAXLS) {
        // restore and either restart from steepest descent or give up
        if (p.col == 0) {
          p.status = 4;
          return;
        }
        p.col = 0;
        p.head = 0;
        p.theta = 1.0;
        begin_iteration(b);  // x, g, f still hold the iterate the search started from
        return;
      }
      set_trial(b);
      return;
    }
    // accepted (CONVERGENCE or WARNING): trial point becomes the new iterate
    std::copy(vx(xtrial, b), vx(xtrial, b) + d, vx(x, b));
    std::copy(gv, gv + d, vx(g, b));
    p.f = fv;
    p.iter++;
    if (p.iter > maxit) {
      p.status = 3;
      return;
    }
    double sb = 0.0;
    for (int i = 0; i < d; ++i) sb = std::max(sb, std::fabs(gv[i]));
    if (sb <= pgtol) {
      p.status = 2;
      return;
    }
    const double ddum = std::max({std::fabs(p.fold), std::fabs(p.f), 1.0});
    if ((p.fold - p.f) <= EPSMCH * factr * ddum) {
      p.status = 1;
      return;
    }
    // correction pair: y = g - g_old, s = stp * d
    double yv[64], sv[64];
    std::vector<double> ybig, sbig;
    double *yy = yv, *ss = sv;
    if (d > 64) { ybig.resize(d); sbig.resize(d); yy = ybig.data(); ss = sbig.data(); }
    const double* go = vx(gold, b);
    for (int i = 0; i < d; ++i) yy[i] = gv[i] - go[i];
    const double rr = dot(yy, yy);
    double dr, dd2;
    if (p.stp == 1.0) {
      dr = p.gd - p.gdold;
      dd2 = -p.gdold;
      for (int i = 0; i < d; ++i) ss[i] = dd[i];
    } else {
      dr = (p.gd - p.gdold) * p.stp;
      dd2 = -p.gdold * p.stp;
      for (int i = 0; i < d; ++i) ss[i] = p.stp * dd[i];
    }
    if (!(dr <= EPSMCH * dd2)) {
      int slot;
      if (p.col < m) {
        slot = (p.head + p.col) % m;
        p.col++;
      } else {
        slot = p.head;
        p.head = (p.head + 1) % m;
      }
      std::copy(ss, ss + d, sp(b, slot));
      std::copy(yy, yy + d, yp(b, slot));
      rho[(size_t)b * m + slot] = 1.0 / dr;
      p.theta = rr / dr;
    }
    begin_iteration(b);
  }
};

extern "C" {

magma_lbfgsb* magma_lbfgsb_create(int B, int d, const double* h_x0, int m, double factr,
                                  double pgtol, int maxit) {
  if (B < 0 || d < 1 || m < 1 || m > 64 || !h_x0) return nullptr;
  magma_lbfgsb* s = new magma_lbfgsb();
  s->B = B;
  s->d = d;
  s->m = m;
  s->maxit = maxit;
  s->factr = factr;
  s->pgtol = pgtol;
  s->pr.resize(B);
  const size_t n = (size_t)B * d;
  s->x.assign(h_x0, h_x0 + n);
  s->g.assign(n, 0.0);
  s->xold.assign(n, 0.0);
  s->gold.assign(n, 0.0);
  s->dir.assign(n, 0.0);
  s->xtrial.assign(h_x0, h_x0 + n);
  s->S.assign(n * m, 0.0);
  s->Y.assign(n * m, 0.0);
  s->rho.assign((size_t)B * m, 0.0);
  s->act.resize(B);
  for (int b = 0; b < B; ++b) s->act[b] = b;
  return s;
}

int magma_lbfgsb_ask(magma_lbfgsb* s, double* h_x, unsigned char* h_active, int* h_nactive) {
  if (!s) return 2;
  int na = 0;
  for (int b = 0; b < s->B; ++b) {
    const bool act = s->pr[b].status == 0;
    h_active[b] = act ? 1 : 0;
    na += act;
    const double* src = act ? s->vx(s->xtrial, b) : s->vx(s->x, b);
    std::copy(src, src + s->d, h_x + (size_t)b * s->d);
  }
  *h_nactive = na;
  return 0;
}

int magma_lbfgsb_tell(magma_lbfgsb* s, const double* h_f, const double* h_g) {
  if (!s) return 2;
  for (int b = 0; b < s->B; ++b)
    if (s->pr[b].status == 0) s->tell_one(b, h_f[b], h_g + (size_t)b * s->d);
  return 0;
}

int magma_lbfgsb_ask_compact(magma_lbfgsb* s, int* h_sel, double* h_x, int* h_nactive) {
  if (!s) return 2;
  size_t k = 0;
  for (size_t q = 0; q < s->act.size(); ++q) {
    const int b = s->act[q];
    if (s->pr[b].status != 0) continue;  // ended by a full-size tell
    s->act[k] = b;
    h_sel[k] = b;
    const double* src = s->vx(s->xtrial, b);
    std::copy(src, src + s->d, h_x + k * s->d);
    ++k;
  }
  s->act.resize(k);
  *h_nactive = (int)k;
  return 0;
}

int magma_lbfgsb_tell_compact(magma_lbfgsb* s, int n, const int* h_sel, const double* h_f,
                              const double* h_g) {
  if (!s) return 2;
  for (int k = 0; k < n; ++k) {
    const int b = h_sel[k];
    if (b < 0 || b >= s->B) return 2;
    if (s->pr[b].status == 0) s->tell_one(b, h_f[k], h_g + (size_t)k * s->d);
  }
  return 0;
}

int magma_lbfgsb_result(magma_lbfgsb* s, double* h_x, double* h_f, int* h_iters, int* h_nfev,
                        int* h_status) {
  if (!s) return 2;
  const double nanv = std::numeric_limits<double>::quiet_NaN();
  for (int b = 0; b < s->B; ++b) {
    const Problem& p = s->pr[b];
    for (int i = 0; i < s->d; ++i)
      h_x[(size_t)b * s->d + i] = (p.status == 5) ? nanv : s->x[(size_t)b * s->d + i];
    if (h_f) h_f[b] = p.f;
    if (h_iters) h_iters[b] = p.iter;
    if (h_nfev) h_nfev[b] = p.nfev;
    if (h_status) h_status[b] = p.status;
  }
  return 0;
}

void magma_lbfgsb_destroy(magma_lbfgsb* s) { delete s; }

}  // extern "C"
