// Batched, lock-step L-BFGS-B (unbounded case) with reverse communication -- host driver.
// The optimiser itself is lbfgsb_core.h (shared with the device-resident M-step kernel); this
// file advances B independent problems together so that one batched objective evaluation (one
// GPU launch over all individuals) serves every problem's current trial point.
#include "../../include/magma_hip.h"
#include "lbfgsb_core.h"

#include <algorithm>
#include <limits>
#include <vector>

namespace {
constexpr int DCAP = 8, MCAP = 8;  // capacity of the host-side problems (the reference uses d <= 3, m = 5)
using Problem = lbcore::State<DCAP, MCAP>;
}  // namespace

struct magma_lbfgsb {
  int B, d, m, maxit;
  double factr, pgtol;
  std::vector<Problem> pr;
  std::vector<int> act;  // problems still running, ascending (compact ask/tell)

  void tell_one(int b, double fv, const double* gv) {
    lbcore::tell(pr[b], fv, gv, factr, pgtol, maxit);
  }
};

extern "C" {

magma_lbfgsb* magma_lbfgsb_create(int B, int d, const double* h_x0, int m, double factr,
                                  double pgtol, int maxit) {
  if (B < 0 || d < 1 || d > DCAP || m < 1 || m > MCAP || !h_x0) return nullptr;
  magma_lbfgsb* s = new magma_lbfgsb();
  s->B = B;
  s->d = d;
  s->m = m;
  s->maxit = maxit;
  s->factr = factr;
  s->pgtol = pgtol;
  s->pr.resize(B);
  for (int b = 0; b < B; ++b) lbcore::init(s->pr[b], d, m, h_x0 + (size_t)b * d);
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
    const double* src = act ? s->pr[b].xtrial : s->pr[b].x;
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
    const double* src = s->pr[b].xtrial;
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
      h_x[(size_t)b * s->d + i] = (p.status == 5) ? nanv : p.x[i];
    if (h_f) h_f[b] = p.f;
    if (h_iters) h_iters[b] = p.iter;
    if (h_nfev) h_nfev[b] = p.nfev;
    if (h_status) h_status[b] = p.status;
  }
  return 0;
}

void magma_lbfgsb_destroy(magma_lbfgsb* s) { delete s; }

}  // extern "C"
