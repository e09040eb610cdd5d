// The EM driver behind the C ABI (include/magma_hip.h, "context / dataset / driver" section):
// `training` (MAGMA.R:15-82), `e_step` / `m_step` / `logL_monitoring` (Computing_functions.R:
// 371-437, 236-264), `posterior_mu` (MAGMA.R:107-133), `train_new_gp` (MAGMA.R:84-104) and
// `pred_gp` against a resident hyper-posterior, as native calls: what an R `.Call` shim, the
// ctypes host or any other FFI reaches with ONE call per reference function, instead of one call
// per kernel. Multi-GPU: one process (or thread) per GPU, each with its own context; partial
// precision sums, weighted means, objective pieces and the sharded theta_0 gradient are combined
// by an all-reduce -- RCCL (loaded at run time, never linked) or a caller-supplied callback.
#include "../../include/magma_hip.h"
#include "common.h"
#include "kernels.h"
#include "driver.h"

#include <rccl/rccl.h>  // types and prototypes only: the library is opened with dlopen
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

using namespace magma;

namespace {

// ------------------------------------------------------------------ RCCL, resolved at run time
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) get_id = nullptr;
  decltype(&ncclCommInitRank) init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclCommDestroy) destroy = nullptr;
  decltype(&ncclGetErrorString) err = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the copy the process already has (PyTorch brings its own) before any other
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.h)
      for (const char* n : names)
        if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.h) return;
    r.get_id = (decltype(r.get_id))dlsym(r.h, "ncclGetUniqueId");
    r.init_rank = (decltype(r.init_rank))dlsym(r.h, "ncclCommInitRank");
    r.all_reduce = (decltype(r.all_reduce))dlsym(r.h, "ncclAllReduce");
    r.destroy = (decltype(r.destroy))dlsym(r.h, "ncclCommDestroy");
    r.err = (decltype(r.err))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.get_id && r.init_rank && r.all_reduce && r.destroy && r.err;
  });
  return r;
}

int nccl_check(ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return 0;
  return fail(200 + (int)e, std::string(what) + ": " + (rccl().err ? rccl().err(e) : "RCCL error"));
}

}  // namespace

// ------------------------------------------------------------------ context
struct magma_ctx {
  int device = 0, rank = 0, world = 1;
  hipStream_t main = nullptr, side = nullptr, low = nullptr, sums = nullptr;
  ncclComm_t comm = nullptr;       // RCCL (world > 1), or
  magma_allreduce_fn cb = nullptr;  // a caller-supplied all-reduce
  void* cb_user = nullptr;
  // compute units kept free of the resident M-step while theta_0 runs (measured at C3: 64 / 80 /
  // 88 / 96 / 104 / 112 -> 119.5 / 119.3 / 118.2 / 121.0 / 120.8 / 118.3 EM it/s, fit 55.7 -> 53.5 ms)
  int reserve_cus = 96;
  int prefactor_wgs = 96;  // workgroups of the theta_0 inverse that runs next to the E-step's
  // workgroups of the monitoring call's inverse while the next E-step's sums run next to it (a
  // tile-graph workgroup fills its compute unit: with the full grid nothing else is dispatched
  // until it is over and the two run one after the other); 0 = full grid
  int monitor_wgs = 160;
  int ncu = 256;

  // sum over ranks, in place, of n doubles in device memory; ordered on `st`
  // a communicator exists (also with world == 1: RCCL with one rank, which the single-GPU tests
  // use to run the whole multi-rank code path)
  bool has_comm() const { return comm != nullptr || cb != nullptr; }

  int allreduce(double* d, int64_t n, hipStream_t st) {
    if (!has_comm() || n == 0) return 0;
    if (cb) {
      // the callback may use its own stream / host staging: hand over finished data
      MAGMA_CHECK_HIP(hipStreamSynchronize(st));
      if (int rc = cb(cb_user, d, n, (void*)st)) return fail(210, "all-reduce callback failed (" + std::to_string(rc) + ")");
      return 0;
    }
    return nccl_check(rccl().all_reduce(d, d, (size_t)n, ncclFloat64, ncclSum, comm, st), "ncclAllReduce");
  }
};

namespace magma {

// ------------------------------------------------------------------ dataset
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int alloc(size_t b) {
    bytes = b;
    return check_hip(hipMalloc(&p, b ? b : 8), "hipMalloc");
  }
  int ensure(size_t b) {  // grow only
    if (p && b <= bytes) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    return alloc(b);
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <class T> T* as() const { return static_cast<T*>(p); }
};

}  // namespace magma

// An individual with more observations than one wavefront handles (n > 64): it takes the dense
// route -- K_i, its inverse and its objective through the N x N kernels on its own timestamps.
struct LargeIndiv {
  int64_t row = 0;  // index among this rank's individuals
  int n = 0, np = 0;
  std::vector<double> h_t, h_y;
  std::vector<int> h_idx;
  DevBuf t, y, idx, off;  // t, y zero padded to np; off = {0, n}; (off2 = {0, 0}: `zero2`)
};

struct magma_dataset {
  magma_ctx* ctx = nullptr;
  // M: the individuals the one-wavefront kernels handle (n <= 64), compact; Mloc: all of this
  // rank's individuals; small_row[k]: index of small individual k among them
  int64_t M = 0, Mloc = 0, M_total = 0, row0 = 0;
  std::vector<int64_t> small_row;
  std::vector<std::unique_ptr<LargeIndiv>> large;
  int np_large = 0;
  DevBuf LK, LKc, LC, Lm, Lwork, zero2, Lscal;
  int nmax = 1, N = 0, Np = 0;
  std::vector<int64_t> h_off, h_off2;
  std::vector<double> h_t, h_y, h_grid;
  std::vector<int> h_idx;
  DevBuf off, off2, t, y, idx, grid;
  DevBuf ptrP, srcP, tgtP, ptrw, srcw, tgtw, Kinv, v;
  int64_t nP = 0, nw = 0;
  DevBuf theta, f, g, info, sel, fsum;
  DevBuf P, w, mean, m0, work, cpack, pack;
  // resident M-step results: x (3 M), f (M), f(x0) (M) doubles, evaluation counts (M ints)
  DevBuf rres, rit, rst, gath, redo_theta;
  void* h_rres = nullptr;  // pinned mirror
  DevBuf plist;  // lower tiles of the theta_0 gradient owned by this rank
  int nplist = 0;
  bool cpack_ok = false, have_order = false;
  bool cpack_early = false;  // set by magma_em_iteration around its E-step
  bool theta_from_estep = false;  // d->theta holds the theta_i rows magma_em_iteration's E-step uploaded
  // d->theta already holds the rows in pre_theta: copied on the device from the resident kernel's
  // results at the end of the EM iteration that produced them (the next E-step uploads nothing)
  bool theta_dev_valid = false;
  std::vector<double> h_m0_last;  // last prior mean uploaded (set_m0)
  double* h_fin = nullptr;        // pinned: (f0, info words) of the E-step's dense part
  double* h_mon = nullptr;        // pinned: (f, g1, g2, info) of a deferred theta_0 evaluation
  int* h_flag = nullptr;          // pinned: per-individual info of the last E-step, counted lazily
  bool flag_pending = false;
  // identical timestamps for all of this rank's individuals (CF:221-224, 347-350): with common
  // hyper-parameters their summed objective is M x one pseudo-individual's (small.hip, dedup_*)
  bool common_times = false, dedup_ok = false;
  // the NEXT E-step's sums over individuals (they depend on theta_i only), started on the side
  // stream next to the monitoring call that ends an EM iteration (prefetch_estep_sums)
  DevBuf P2, w2, theta2, info2, f2;
  DevBuf covfac;  // V = L^-1 of the last E-step's precision (cov = V^T V); see driver.h
  bool covfac_ok = false;
  // the theta_0 chain's own workspace, and the inverse of its first evaluation started early
  // (theta0_prefactor_impl, driver.h)
  DevBuf work0;
  hipEvent_t pre0_ev = nullptr;
  bool pre0_issued = false;
  double pre0_key[3] = {0, 0, 0};
  hipEvent_t pre_ev = nullptr;
  bool pre_valid = false, pre_issued = false;
  std::vector<double> pre_theta;
  // ... started even earlier inside an EM iteration: right behind the resident M-step kernel, from
  // the theta_i it leaves in device memory (want_early: set by magma_em_iteration around its
  // M-step; pre_early: issued; m_redo: problems were solved again on the host, the device rows are
  // not the final ones)
  bool want_early = false, pre_early = false, m_redo = false;
  hipEvent_t res_ev = nullptr;
  // The NEXT E-step's dense part, queued behind the monitoring evaluation that ends an EM
  // iteration (all of its inputs are known then): precision + K_0^-1 -> cov in P2, mean in mean2,
  // packed cov blocks and factor of cov for the next M-step. Taken by the next magma_e_step when
  // it is called with exactly these hyper-parameters and prior mean; the current mean / cov
  // (`param` of a fit that stops here, MG:78) stay where they are.
  DevBuf mean2;
  double* h_fin2 = nullptr;
  hipEvent_t mon_ev = nullptr, spec_ev = nullptr;
  bool spec_issued = false, spec_cpack = false, spec_fac = false;
  double spec_key[3] = {0, 0, 0};
  std::vector<double> spec_m0;
  ~magma_dataset() {
    if (res_ev) (void)hipEventDestroy(res_ev);
    if (mon_ev) (void)hipEventDestroy(mon_ev);
    if (spec_ev) (void)hipEventDestroy(spec_ev);
    if (h_fin2) (void)hipHostFree(h_fin2);
    if (pre_ev) (void)hipEventDestroy(pre_ev);
    if (pre0_ev) (void)hipEventDestroy(pre0_ev);
    if (h_rres) (void)hipHostFree(h_rres);
    if (h_mon) (void)hipHostFree(h_mon);
    if (h_fin) (void)hipHostFree(h_fin);
    if (h_flag) (void)hipHostFree(h_flag);
  }
  DevBuf ddS, ddC, ddy, ddoff, ddf, ddg, ddinfo;
  bool a0_valid = false;
  double a0_key[3] = {0, 0, 0};
  std::vector<int> h_perm;
  DevBuf order;
  // by-products of the last E-step / M-step (logL_monitoring shortcuts, MG:54,60)
  double last_f0 = 0.0, last_sum_fi = 0.0, last_sum_fi_new = 0.0;
  magma_em_stats stats = {0, 0, 0, 0, 0, 0};
};

namespace {

// Host-side phase timing of the driver (MAGMA_HOST_TIMING=1: printed when a context is destroyed).
struct HostTiming {
  bool on = getenv("MAGMA_HOST_TIMING") != nullptr;
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  std::map<std::string, std::pair<long, double>> acc;
  void mark(const char* label) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    auto& a = acc[label];
    a.first += 1;
    a.second += std::chrono::duration<double, std::micro>(now - last).count();
    last = now;
  }
  void dump() {
    if (!on) return;
    for (auto& kv : acc)
      fprintf(stderr, "[magma host timing] %-28s n=%6ld  avg %9.1f us  total %10.1f ms\n",
              kv.first.c_str(), kv.second.first, kv.second.second / kv.second.first, kv.second.second * 1e-3);
    acc.clear();
  }
};
HostTiming g_ht;
double g_ht_prev_join = 0.0;
#define HT(label) g_ht.mark(label)

struct Opt {
  int m = 5, maxit = 100;
  double factr = 1e7, pgtol = 0.0;
  explicit Opt(const double* o) {
    if (o) {
      m = (int)o[0];
      factr = o[1];
      pgtol = o[2];
      maxit = (int)o[3];
    }
  }
};

int set_device(magma_ctx* c) { return check_hip(hipSetDevice(c->device), "hipSetDevice"); }

template <class T>
int upload(DevBuf& b, const std::vector<T>& h) {
  if (int rc = b.alloc(h.size() * sizeof(T))) return rc;
  if (h.empty()) return 0;
  return check_hip(hipMemcpy(b.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice), "upload");
}

// Source lists of the deterministic E-step sums (magma_estep_gather): for every element of the
// lower triangle of P (of w) that receives anything, the positions of its terms in the
// materialised K_i^-1 (in v), individuals in order, each one's lower triangle column by column --
// a counting sort by target, so the order of summation is fixed by the dataset alone.
int build_gather_lists(magma_dataset* d) {
  const int64_t M = d->M;
  const int64_t Np = d->Np;
  int64_t npairs = 0;
  for (int64_t i = 0; i < M; ++i) {
    const int64_t n = d->h_off[i + 1] - d->h_off[i];
    npairs += n * (n + 1) / 2;
  }
  if (d->h_off2[M] >= ((int64_t)1 << 31))
    return fail(2, "dataset: more than 2^31 matrix entries on one rank (use more ranks)");
  // pass 1: counts per target (column-major position in P)
  std::vector<int> cnt((size_t)Np * Np + 1, 0);
  auto for_pairs = [&](auto&& fn) {
    for (int64_t i = 0; i < M; ++i) {
      const int64_t o = d->h_off[i], n = d->h_off[i + 1] - o, o2 = d->h_off2[i];
      const int* ix = d->h_idx.data() + o;
      for (int64_t c = 0; c < n; ++c)
        for (int64_t r = c; r < n; ++r)  // r >= c: idx ascending, so the lower triangle of P
          fn((int64_t)ix[r] + (int64_t)ix[c] * Np, (int)(o2 + c * n + r));
    }
  };
  for_pairs([&](int64_t tg, int) { ++cnt[(size_t)tg]; });
  std::vector<int64_t> tgt, ptr;
  std::vector<int64_t> start((size_t)Np * Np, -1);
  int64_t acc = 0;
  for (int64_t e = 0; e < Np * Np; ++e)
    if (cnt[(size_t)e]) {
      start[(size_t)e] = acc;
      tgt.push_back(e);
      ptr.push_back(acc);
      acc += cnt[(size_t)e];
    }
  ptr.push_back(acc);
  std::vector<int> src((size_t)npairs);
  for_pairs([&](int64_t tg, int s) { src[(size_t)start[(size_t)tg]++] = s; });
  d->nP = (int64_t)tgt.size();
  if (int rc = upload(d->ptrP, ptr)) return rc;
  if (int rc = upload(d->srcP, src)) return rc;
  if (int rc = upload(d->tgtP, tgt)) return rc;
  // w: sources = positions in v (= rows of the ragged arrays), grouped by grid point
  std::vector<int> cw((size_t)Np, 0);
  const int64_t R = d->h_off[M];
  for (int64_t k = 0; k < R; ++k) ++cw[(size_t)d->h_idx[(size_t)k]];
  std::vector<int64_t> tw, pw, sw_start((size_t)Np, -1);
  acc = 0;
  for (int64_t e = 0; e < Np; ++e)
    if (cw[(size_t)e]) {
      sw_start[(size_t)e] = acc;
      tw.push_back(e);
      pw.push_back(acc);
      acc += cw[(size_t)e];
    }
  pw.push_back(acc);
  std::vector<int> sw((size_t)R);
  for (int64_t k = 0; k < R; ++k) sw[(size_t)sw_start[(size_t)d->h_idx[(size_t)k]]++] = (int)k;
  d->nw = (int64_t)tw.size();
  if (int rc = upload(d->ptrw, pw)) return rc;
  if (int rc = upload(d->srcw, sw)) return rc;
  return upload(d->tgtw, tw);
}

int h2d(void* dst, const void* src, size_t bytes, hipStream_t st) {
  return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st), "h2d");
}
int d2h(void* dst, const void* src, size_t bytes, hipStream_t st) {
  return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st), "d2h");
}
int sync(hipStream_t st) { return check_hip(hipStreamSynchronize(st), "stream sync"); }

// theta_i rows of this rank -> d->theta; returns the stride for the kernels (3, or 0 = common)
int set_theta(magma_dataset* d, const double* h_theta_i, int stride, hipStream_t st) {
  d->theta_dev_valid = false;
  if (stride == 0) return h2d(d->theta.p, h_theta_i, 3 * sizeof(double), st);
  if (d->M == 0) return 0;
  if (d->large.empty())
    return h2d(d->theta.p, h_theta_i + 3 * d->row0, (size_t)(3 * d->M) * sizeof(double), st);
  std::vector<double> rows((size_t)(3 * d->M));
  for (int64_t k = 0; k < d->M; ++k)
    for (int q = 0; q < 3; ++q) rows[(size_t)(3 * k + q)] = h_theta_i[3 * (d->row0 + d->small_row[(size_t)k]) + q];
  if (int rc = h2d(d->theta.p, rows.data(), rows.size() * sizeof(double), st)) return rc;
  return sync(st);  // rows goes out of scope
}

// theta of local individual `row` (stride 0: the common triple)
const double* theta_row(const magma_dataset* d, const double* h_theta_i, int stride, int64_t row) {
  return stride == 0 ? h_theta_i : h_theta_i + 3 * (d->row0 + row);
}

// One large individual's f (and gradient) at th3 against the resident hyper-posterior.
int large_eval(magma_ctx* c, magma_dataset* d, LargeIndiv& L, const double* th3, int want_grad,
               double* out4, int* info) {
  hipStream_t st = c->main;
  if (int rc = launch_gather_block(L.n, L.np, L.idx.as<int>(), d->P.as<double>(), d->Np,
                                   d->mean.as<double>(), d->LC.as<double>(), d->Lm.as<double>(), st))
    return rc;
  if (want_grad) d->stats.indiv_evals += 1;
  return indiv_dense_eval_impl(L.n, L.t.as<double>(), th3, L.y.as<double>(), d->Lm.as<double>(),
                               d->LC.as<double>(), want_grad, d->Lwork.p, out4, info, st);
}

int set_m0(magma_dataset* d, const double* h_m0, int n_m0, hipStream_t st) {
  std::vector<double> v((size_t)d->Np, 0.0);
  if (n_m0 != 1 && n_m0 != d->N) return fail(2, "m_0 must be a scalar or one value per grid point");
  for (int k = 0; k < d->N; ++k) v[(size_t)k] = n_m0 == 1 ? h_m0[0] : h_m0[k];
  if (v == d->h_m0_last) return 0;  // (every E-step of a fit passes the same prior mean)
  d->h_m0_last = v;
  MAGMA_CHECK_HIP(hipMemcpyAsync(d->m0.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, st));
  return sync(st);  // v goes out of scope
}

double mean_sigma(const magma_dataset* d, const double* th, int stride, double scale) {
  if (stride == 0) return scale * th[2];
  double s = 0.0;
  for (int64_t i = 0; i < d->M_total; ++i) s += scale * th[3 * i + 2];
  return s / (double)d->M_total;
}

// the per-individual info words of the last E-step, copied to pinned memory on st: counted once st
// has been synchronised (do_sync = false: the caller just did)
int count_flagged(magma_dataset* d, hipStream_t st, bool do_sync = true) {
  if (!d->flag_pending) return 0;
  if (do_sync)
    if (int rc = sync(st)) return rc;
  for (int64_t k = 0; k < d->M; ++k) d->stats.flagged += (d->h_flag[k] != 0);
  d->flag_pending = false;
  return 0;
}

int ensure_cpack(magma_dataset* d, hipStream_t st) {
  if (d->cpack_ok || d->M == 0) return 0;
  if (int rc = magma_indiv_gather_cov((int)d->M, d->nmax, d->off.as<int64_t>(), d->idx.as<int>(),
                                      d->P.as<double>(), d->N, d->cpack.as<double>(), st))
    return rc;
  d->cpack_ok = true;
  return 0;
}

// The sums over individuals of the next E-step (CF:382-391: sum_i scatter K_i^-1, sum_i K_i^-1 y_i)
// depend on theta_i alone: an EM iteration starts them on the side stream as soon as its M-step
// has the new theta_i, next to the (latency-bound) dense inverse of its last monitoring call,
// into a second pair of buffers. The E-step that follows takes them if it is called with these
// very hyper-parameters (compared value by value), else computes its own.
int prefetch_estep_sums(magma_ctx* c, magma_dataset* d, const double* h_theta_i) {
  d->pre_valid = false;
  if (d->M == 0 || !prefetch_enabled()) return 0;
  hipStream_t st = c->side;
  if (d->pre_issued)  // (sums started behind the resident kernel that turned out not to be final)
    MAGMA_CHECK_HIP(hipStreamWaitEvent(st, d->pre_ev, 0));
  std::vector<double> rows((size_t)(3 * d->M));
  for (int64_t k = 0; k < d->M; ++k)
    for (int q = 0; q < 3; ++q)
      rows[(size_t)(3 * k + q)] = h_theta_i[3 * (d->row0 + d->small_row[(size_t)k]) + q];
  // (pageable source: the copy has left `rows` when the call returns)
  MAGMA_CHECK_HIP(hipMemcpyAsync(d->theta2.p, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, st));
  const size_t Np = (size_t)d->Np;
  MAGMA_CHECK_HIP(hipMemsetAsync(d->P2.p, 0, sizeof(double) * Np * Np, st));
  MAGMA_CHECK_HIP(hipMemsetAsync(d->w2.p, 0, sizeof(double) * Np, st));
  d->pre_issued = true;
  if (int rc = magma_estep_gather((int)d->M, d->nmax, d->off.as<int64_t>(), d->off2.as<int64_t>(),
                                  d->t.as<double>(), d->y.as<double>(), d->theta2.as<double>(), 3,
                                  d->Kinv.as<double>(), d->v.as<double>(), d->f2.as<double>(), d->nP,
                                  d->ptrP.as<int64_t>(), d->srcP.as<int>(), d->tgtP.as<int64_t>(), d->nw,
                                  d->ptrw.as<int64_t>(), d->srcw.as<int>(), d->tgtw.as<int64_t>(),
                                  d->P2.as<double>(), d->w2.as<double>(), d->info2.as<int>(), st))
    return rc;
  MAGMA_CHECK_HIP(hipEventRecord(d->pre_ev, st));
  d->pre_theta.assign(h_theta_i, h_theta_i + 3 * d->M_total);
  d->pre_valid = true;
  return 0;
}

// The same sums started even earlier: on the `sums` stream right behind the resident M-step
// kernel, from the theta_i rows that kernel leaves in device memory (no host round trip, no
// second upload). The caller validates them once the M-step's results are on the host.
int early_estep_sums(magma_ctx* c, magma_dataset* d) {
  if (d->M == 0 || !prefetch_enabled() || !d->res_ev) return 0;
  hipStream_t st = c->sums;
  MAGMA_CHECK_HIP(hipEventRecord(d->res_ev, c->main));
  MAGMA_CHECK_HIP(hipStreamWaitEvent(st, d->res_ev, 0));
  if (d->pre_issued) MAGMA_CHECK_HIP(hipStreamWaitEvent(st, d->pre_ev, 0));
  const size_t Np = (size_t)d->Np;
  MAGMA_CHECK_HIP(hipMemsetAsync(d->P2.p, 0, sizeof(double) * Np * Np, st));
  MAGMA_CHECK_HIP(hipMemsetAsync(d->w2.p, 0, sizeof(double) * Np, st));
  d->pre_issued = true;
  d->pre_valid = false;
  if (int rc = magma_estep_gather((int)d->M, d->nmax, d->off.as<int64_t>(), d->off2.as<int64_t>(),
                                  d->t.as<double>(), d->y.as<double>(), d->rres.as<double>(), 3,
                                  d->Kinv.as<double>(), d->v.as<double>(), d->f2.as<double>(), d->nP,
                                  d->ptrP.as<int64_t>(), d->srcP.as<int>(), d->tgtP.as<int64_t>(), d->nw,
                                  d->ptrw.as<int64_t>(), d->srcw.as<int>(), d->tgtw.as<int64_t>(),
                                  d->P2.as<double>(), d->w2.as<double>(), d->info2.as<int>(), st))
    return rc;
  MAGMA_CHECK_HIP(hipEventRecord(d->pre_ev, st));
  d->pre_early = true;
  return 0;
}

std::atomic<bool> g_speculate{true};

// The next E-step's dense part queued on the main stream behind whatever is there (the monitoring
// evaluation, whose K_0^-1(theta_0, pen) in d->work it reuses). See magma_dataset::spec_*.
int speculate_estep(magma_ctx* c, magma_dataset* d, const double* h_theta0, double pen, bool want_cpack) {
  d->spec_issued = false;
  if (!g_speculate.load() || c->has_comm() || !d->large.empty() || !d->pre_valid || !d->spec_ev || d->M == 0)
    return 0;
  hipStream_t st = c->main;
  MAGMA_CHECK_HIP(hipStreamWaitEvent(st, d->pre_ev, 0));
  d->pre_issued = false;  // (consumed: the sums in P2 become the speculative cov in place)
  d->pre_valid = false;
  if (int rc = estep_finish_launch(d->N, d->grid.as<double>(), h_theta0[0], h_theta0[1], pen,
                                   d->m0.as<double>(), d->P2.as<double>(), d->w2.as<double>(),
                                   d->mean2.as<double>(), d->work.p, 1, 1, d->h_fin2, st))
    return rc;
  // the packed blocks and the factor now describe the speculative cov, not the current one
  d->cpack_ok = false;
  d->covfac_ok = false;
  d->spec_cpack = want_cpack;
  if (want_cpack)
    if (int rc = magma_indiv_gather_cov((int)d->M, d->nmax, d->off.as<int64_t>(), d->idx.as<int>(),
                                        d->P2.as<double>(), d->N, d->cpack.as<double>(), st))
      return rc;
  d->spec_fac = d->covfac.p && covfac_enabled();
  if (d->spec_fac)
    if (int rc = estep_cov_factor(d->N, d->work.p, d->covfac.as<double>(), st)) return rc;
  if (!d->h_flag)
    MAGMA_CHECK_HIP(hipHostMalloc((void**)&d->h_flag, sizeof(int) * (size_t)d->M, hipHostMallocDefault));
  MAGMA_CHECK_HIP(hipMemcpyAsync(d->h_flag, d->info2.p, sizeof(int) * (size_t)d->M, hipMemcpyDeviceToHost, st));
  MAGMA_CHECK_HIP(hipEventRecord(d->spec_ev, st));
  d->spec_key[0] = h_theta0[0];
  d->spec_key[1] = h_theta0[1];
  d->spec_key[2] = pen;
  d->spec_m0 = d->h_m0_last;
  d->spec_issued = true;
  return 0;
}

// sum over this rank's individuals of f (and of the three gradient components) -> out4 (host),
// already summed over ranks
int sum_fi(magma_ctx* c, magma_dataset* d, const double* h_theta_i, int stride, int want_grad,
           double* out4) {
  hipStream_t st = c->main;
  double* fs = d->fsum.as<double>();
  MAGMA_CHECK_HIP(hipMemsetAsync(fs, 0, 4 * sizeof(double), st));
  if (stride == 0 && d->common_times && dedup_enabled()) {
    // one pseudo-individual stands for all M (see dedup_outer_kernel): its f, g times M
    const int n0 = (int)d->h_off[1];
    if (int rc = set_theta(d, h_theta_i, 0, st)) return rc;
    if (!d->dedup_ok) {
      if (int rc = launch_dedup_prepare((int)d->M, n0, d->nmax, d->y.as<double>(), d->idx.as<int>(),
                                        d->mean.as<double>(), d->P.as<double>(), d->Np,
                                        d->ddS.as<double>(), d->ddC.as<double>(), d->ddy.as<double>(), st))
        return rc;
      d->dedup_ok = true;
    }
    if (int rc = magma_indiv_eval(0, 1, d->nmax, d->ddoff.as<int64_t>(), d->t.as<double>(),
                                  d->ddy.as<double>(), d->idx.as<int>(), d->theta.as<double>(), 0,
                                  d->mean.as<double>(), d->ddC.as<double>(), want_grad, nullptr, 0,
                                  d->ddf.as<double>(), d->ddg.as<double>(), d->ddinfo.as<int>(), st))
      return rc;
    if (want_grad) d->stats.indiv_evals += 1;
    double loc[4] = {0.0, 0.0, 0.0, 0.0};
    if (int rc = d2h(loc, d->ddf.p, sizeof(double), st)) return rc;
    if (want_grad)
      if (int rc = d2h(loc + 1, d->ddg.p, 3 * sizeof(double), st)) return rc;
    if (int rc = sync(st)) return rc;
    for (int q = 0; q < 4; ++q) loc[q] *= (double)d->M;
    if (!c->has_comm()) {
      std::copy(loc, loc + 4, out4);
      return 0;
    }
    if (int rc = h2d(fs, loc, 4 * sizeof(double), st)) return rc;
    if (int rc = sync(st)) return rc;
  } else if (d->M > 0) {
    if (int rc = set_theta(d, h_theta_i, stride, st)) return rc;
    if (int rc = ensure_cpack(d, st)) return rc;
    if (int rc = magma_indiv_eval(0, (int)d->M, d->nmax, d->off.as<int64_t>(), d->t.as<double>(),
                                  d->y.as<double>(), d->idx.as<int>(), d->theta.as<double>(), stride,
                                  d->mean.as<double>(), d->cpack.as<double>(), want_grad, nullptr, 0,
                                  d->f.as<double>(), d->g.as<double>(), d->info.as<int>(), st))
      return rc;
    if (want_grad) d->stats.indiv_evals += d->M;
    if (int rc = launch_sum4(d->M, d->f.as<double>(), want_grad ? d->g.as<double>() : nullptr, fs, st))
      return rc;
  }
  if (!d->large.empty()) {  // dense route, one by one; added on the host, then back for the all-reduce
    double loc[4];
    if (int rc = d2h(loc, fs, 4 * sizeof(double), st)) return rc;
    if (int rc = sync(st)) return rc;
    for (auto& L : d->large) {
      double o4[4];
      int info = 0;
      if (int rc = large_eval(c, d, *L, theta_row(d, h_theta_i, stride, L->row), want_grad, o4, &info)) return rc;
      for (int q = 0; q < 4; ++q) loc[q] += o4[q];
    }
    if (int rc = h2d(fs, loc, 4 * sizeof(double), st)) return rc;
    if (int rc = sync(st)) return rc;
  }
  if (int rc = c->allreduce(fs, 4, st)) return rc;
  if (int rc = d2h(out4, fs, 4 * sizeof(double), st)) return rc;
  return sync(st);
}

int theta0_f(magma_ctx* c, magma_dataset* d, const double* th0, double pen, double* f_out) {
  double out3[3];
  int info = 0;
  d->stats.theta0_evals += 1;
  if (int rc = magma_theta0_eval(d->N, d->grid.as<double>(), th0[0], th0[1], pen, d->mean.as<double>(),
                                 d->m0.as<double>(), d->P.as<double>(), 0, d->work.p, out3, &info,
                                 c->main))
    return rc;
  d->a0_valid = (info == 0);
  d->a0_key[0] = th0[0];
  d->a0_key[1] = th0[1];
  d->a0_key[2] = pen;
  *f_out = info == 0 ? out3[0] : std::numeric_limits<double>::quiet_NaN();
  return 0;
}

}  // namespace

namespace magma {
void set_speculate(bool on) { g_speculate.store(on); }
}  // namespace magma

// =================================================================== C ABI
extern "C" {

int magma_comm_unique_id(void* out128) {
  if (!out128) return fail(2, "unique id: NULL buffer");
  if (!rccl().ok) return fail(4, "RCCL is not available in this process (librccl.so could not be opened)");
  ncclUniqueId id;
  if (int rc = nccl_check(rccl().get_id(&id), "ncclGetUniqueId")) return rc;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

static int ctx_common(magma_ctx* c, int device) {
  c->device = device;
  if (int rc = set_device(c)) return rc;
  int lo = 0, hi = 0;
  MAGMA_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  MAGMA_CHECK_HIP(hipStreamCreateWithFlags(&c->main, hipStreamNonBlocking));
  // high priority: the theta_0 chain is a long sequence of small dependent kernels that should
  // not queue behind the wide per-individual launches it overlaps with
  MAGMA_CHECK_HIP(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi));
  MAGMA_CHECK_HIP(hipStreamCreateWithPriority(&c->low, hipStreamNonBlocking, lo));
  // the next E-step's sums when they are started right behind the resident M-step kernel
  // (its priority makes no measurable difference next to the monitoring inverse)
  MAGMA_CHECK_HIP(hipStreamCreateWithFlags(&c->sums, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  MAGMA_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  c->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("MAGMA_RESERVE_CUS")) c->reserve_cus = atoi(e);
  if (const char* e = getenv("MAGMA_PREFACTOR_WGS")) c->prefactor_wgs = atoi(e);
  if (const char* e = getenv("MAGMA_MONITOR_WGS")) c->monitor_wgs = atoi(e);
  return 0;
}

int magma_ctx_create(magma_ctx** out, int device) {
  if (!out) return fail(2, "ctx: NULL output");
  magma_ctx* c = new magma_ctx();
  if (int rc = ctx_common(c, device)) {
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

int magma_ctx_create_rccl(magma_ctx** out, int device, int rank, int world, const void* unique_id128) {
  if (!out || !unique_id128) return fail(2, "ctx: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(2, "ctx: rank must be in 0..world-1");
  if (!rccl().ok) return fail(4, "RCCL is not available in this process (librccl.so could not be opened)");
  magma_ctx* c = new magma_ctx();
  if (int rc = ctx_common(c, device)) {
    delete c;
    return rc;
  }
  c->rank = rank;
  c->world = world;
  {
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    if (int rc = nccl_check(rccl().init_rank(&c->comm, world, id, rank), "ncclCommInitRank")) {
      delete c;
      return rc;
    }
  }
  *out = c;
  return 0;
}

int magma_ctx_create_callback(magma_ctx** out, int device, int rank, int world,
                              magma_allreduce_fn fn, void* user) {
  if (!out) return fail(2, "ctx: NULL output");
  if (world < 1 || rank < 0 || rank >= world) return fail(2, "ctx: rank must be in 0..world-1");
  if (!fn) return fail(2, "ctx: an all-reduce callback is required");
  magma_ctx* c = new magma_ctx();
  if (int rc = ctx_common(c, device)) {
    delete c;
    return rc;
  }
  c->rank = rank;
  c->world = world;
  c->cb = fn;
  c->cb_user = user;
  *out = c;
  return 0;
}

void magma_ctx_destroy(magma_ctx* c) {
  g_ht.dump();
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl().ok) (void)rccl().destroy(c->comm);
  if (c->main) (void)hipStreamDestroy(c->main);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->low) (void)hipStreamDestroy(c->low);
  if (c->sums) (void)hipStreamDestroy(c->sums);
  delete c;
}

int magma_ctx_info(const magma_ctx* c, int* rank, int* world, int* device) {
  if (!c) return fail(2, "ctx is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (device) *device = c->device;
  return 0;
}

void* magma_ctx_stream(const magma_ctx* c) { return c ? (void*)c->main : nullptr; }

// A known vector through the context's communicator on both of the library's streams, before a
// job trusts it with an EM: every rank contributes (rank + 1) x {1, 2, ..., 64} and must get
// world (world + 1) / 2 times that back. RCCL collectives are asynchronous: the streams are polled
// against the deadline instead of synchronised, so a collective that never completes (a rank that
// did not join, a transport that is not up) comes back as status 211 with the process still in
// control (the stream stays blocked: the caller should report and exit, not go on).
int magma_ctx_selfcheck(magma_ctx* c, int timeout_ms) {
  if (!c) return fail(2, "ctx is NULL");
  if (!c->has_comm()) return 0;
  if (int rc = set_device(c)) return rc;
  constexpr int NV = 64;
  std::vector<double> h((size_t)NV);
  for (int k = 0; k < NV; ++k) h[(size_t)k] = (double)(c->rank + 1) * (double)(k + 1);
  const double want = 0.5 * (double)c->world * (double)(c->world + 1);
  hipStream_t streams[2] = {c->main, c->side};
  const char* names[2] = {"main", "side"};
  for (int q = 0; q < 2; ++q) {
    DevBuf b;
    if (int rc = upload(b, h)) return rc;
    if (int rc = c->allreduce(b.as<double>(), NV, streams[q])) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(streams[q]);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) return check_hip(e, "selfcheck stream query");
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms > (double)timeout_ms) {
        b.p = nullptr;  // (the collective may still write into it: leaked on purpose)
        return fail(211, std::string("communicator self-check: the all-reduce on the ") + names[q] +
                             " stream did not complete within " + std::to_string(timeout_ms) + " ms");
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    std::vector<double> r((size_t)NV);
    MAGMA_CHECK_HIP(hipMemcpy(r.data(), b.p, sizeof(double) * NV, hipMemcpyDeviceToHost));
    for (int k = 0; k < NV; ++k)
      if (r[(size_t)k] != want * (double)(k + 1))
        return fail(212, std::string("communicator self-check: wrong sum on the ") + names[q] + " stream (element " +
                             std::to_string(k) + ": " + std::to_string(r[(size_t)k]) + ", expected " +
                             std::to_string(want * (double)(k + 1)) + ")");
  }
  return 0;
}

// ------------------------------------------------------------------ dataset
int magma_dataset_create(magma_ctx* c, magma_dataset** out, int64_t M, int64_t M_total, int64_t row0,
                         const int64_t* h_off, const double* h_t, const double* h_y, int N,
                         const double* h_grid) {
  if (!c || !out) return fail(2, "dataset: NULL argument");
  if (M < 0 || M_total < M || row0 < 0 || row0 + M > M_total) return fail(2, "dataset: bad shard bounds");
  if (N < 1 || !h_grid) return fail(2, "dataset: the union grid needs at least one point");
  if (M > 0 && (!h_off || !h_t || !h_y)) return fail(2, "dataset: NULL arrays");
  if (int rc = set_device(c)) return rc;
  for (int k = 1; k < N; ++k)
    if (!(h_grid[k] > h_grid[k - 1])) return fail(2, "dataset: the grid must be strictly ascending");
  magma_dataset* d = new magma_dataset();
  std::unique_ptr<magma_dataset> guard(d);
  d->ctx = c;
  d->Mloc = M;
  d->M_total = M_total;
  d->row0 = row0;
  d->N = N;
  d->Np = magma_pad(N);
  if (M > 0 && h_off[0] != 0) return fail(2, "dataset: off[0] must be 0");
  d->h_grid.assign((size_t)d->Np, 0.0);
  std::copy(h_grid, h_grid + N, d->h_grid.begin());
  d->h_off.assign(1, 0);
  d->h_off2.assign(1, 0);
  d->nmax = 1;
  int nmaxL = 0;
  for (int64_t i = 0; i < M; ++i) {
    const int64_t o = h_off[i], n = h_off[i + 1] - o;
    if (n < 1) return fail(2, "dataset: every individual needs at least one observation");
    std::vector<int> ix((size_t)n);
    for (int64_t k = o; k < o + n; ++k) {
      if (k > o && !(h_t[k] > h_t[k - 1]))
        return fail(2, "dataset: timestamps must be strictly ascending within an individual");
      // exact match on the grid: replaces the reference's `paste0('X', t)` names (CF:78,451)
      const double* p = std::lower_bound(h_grid, h_grid + N, h_t[k]);
      if (p == h_grid + N || *p != h_t[k]) return fail(2, "dataset: a timestamp is not on the grid");
      ix[(size_t)(k - o)] = (int)(p - h_grid);
    }
    if (n <= MAGMA_NSMALL_MAX) {  // one wavefront per individual
      d->small_row.push_back(i);
      d->nmax = std::max(d->nmax, (int)n);
      d->h_t.insert(d->h_t.end(), h_t + o, h_t + o + n);
      d->h_y.insert(d->h_y.end(), h_y + o, h_y + o + n);
      d->h_idx.insert(d->h_idx.end(), ix.begin(), ix.end());
      d->h_off.push_back(d->h_off.back() + n);
      d->h_off2.push_back(d->h_off2.back() + n * n);
    } else {  // the dense route
      std::unique_ptr<LargeIndiv> L(new LargeIndiv());
      L->row = i;
      L->n = (int)n;
      L->np = magma_pad((int)n);
      L->h_t.assign((size_t)L->np, 0.0);
      L->h_y.assign((size_t)L->np, 0.0);
      std::copy(h_t + o, h_t + o + n, L->h_t.begin());
      std::copy(h_y + o, h_y + o + n, L->h_y.begin());
      L->h_idx = ix;
      nmaxL = std::max(nmaxL, (int)n);
      d->large.push_back(std::move(L));
    }
  }
  d->M = (int64_t)d->small_row.size();
  M = d->M;  // from here on: the compact set of small individuals
  const int64_t R = d->h_off[(size_t)M];
  int rc;
  if ((rc = upload(d->off, d->h_off)) || (rc = upload(d->off2, d->h_off2)) || (rc = upload(d->t, d->h_t)) ||
      (rc = upload(d->y, d->h_y)) || (rc = upload(d->idx, d->h_idx)) || (rc = upload(d->grid, d->h_grid)))
    return rc;
  if ((rc = build_gather_lists(d))) return rc;
  d->common_times = d->large.empty() && M > 1;
  for (int64_t i = 1; i < M && d->common_times; ++i) {
    const int64_t n0 = d->h_off[1], o = d->h_off[(size_t)i];
    d->common_times = (d->h_off[(size_t)i + 1] - o == n0) &&
                      std::equal(d->h_t.begin(), d->h_t.begin() + n0, d->h_t.begin() + o);
  }
  if (d->common_times) {
    const size_t n0 = (size_t)d->h_off[1];
    const std::vector<int64_t> o1 = {0, (int64_t)n0};
    if ((rc = d->ddS.alloc(sizeof(double) * n0 * n0)) ||
        (rc = d->ddC.alloc(sizeof(double) * (size_t)magma_cpack_stride(d->nmax))) ||
        (rc = d->ddy.alloc(sizeof(double) * n0)) || (rc = upload(d->ddoff, o1)) ||
        (rc = d->ddf.alloc(sizeof(double))) || (rc = d->ddg.alloc(sizeof(double) * 3)) ||
        (rc = d->ddinfo.alloc(sizeof(int))))
      return rc;
  }
  const size_t Mb = (size_t)std::max<int64_t>(M, 1), Np = (size_t)d->Np;
  const size_t Mt = (size_t)std::max<int64_t>(M_total, 1);
  const int nt = d->Np / 64;
  if ((rc = d->Kinv.alloc(sizeof(double) * (size_t)std::max<int64_t>(d->h_off2[(size_t)M], 1))) ||
      (rc = d->v.alloc(sizeof(double) * (size_t)std::max<int64_t>(R, 1))) ||
      (rc = d->theta.alloc(sizeof(double) * 3 * Mb)) || (rc = d->f.alloc(sizeof(double) * Mb)) ||
      (rc = d->g.alloc(sizeof(double) * 3 * Mb)) || (rc = d->info.alloc(sizeof(int) * Mb)) ||
      (rc = d->sel.alloc(sizeof(int) * Mb)) || (rc = d->order.alloc(sizeof(int) * Mb)) || (rc = d->fsum.alloc(sizeof(double) * 16)) ||
      (rc = d->P.alloc(sizeof(double) * Np * Np)) || (rc = d->w.alloc(sizeof(double) * Np)) ||
      (rc = d->mean.alloc(sizeof(double) * Np)) || (rc = d->m0.alloc(sizeof(double) * Np)) ||
      (rc = d->work.alloc(magma_workspace_bytes(N))) || (rc = d->work0.alloc(magma_workspace_bytes(N))) ||
      (rc = d->cpack.alloc(sizeof(double) * Mb * (size_t)magma_cpack_stride(d->nmax))) ||
      (rc = d->rres.alloc(sizeof(double) * 6 * Mb)) || (rc = d->rit.alloc(sizeof(int) * Mb)) ||
      (rc = d->rst.alloc(sizeof(int) * Mb)))
    return rc;
  if (M > 0) {
    if ((rc = d->P2.alloc(sizeof(double) * Np * Np)) || (rc = d->w2.alloc(sizeof(double) * Np)) ||
        (rc = d->covfac.alloc(sizeof(double) * Np * Np)) ||
        (rc = d->theta2.alloc(sizeof(double) * 3 * Mb)) || (rc = d->info2.alloc(sizeof(int) * Mb)) ||
        (rc = d->f2.alloc(sizeof(double) * Mb)) || (rc = d->mean2.alloc(sizeof(double) * Np)))
      return rc;
    MAGMA_CHECK_HIP(hipEventCreateWithFlags(&d->res_ev, hipEventDisableTiming));
    MAGMA_CHECK_HIP(hipEventCreateWithFlags(&d->mon_ev, hipEventDisableTiming));
    MAGMA_CHECK_HIP(hipEventCreateWithFlags(&d->spec_ev, hipEventDisableTiming));
    MAGMA_CHECK_HIP(hipHostMalloc((void**)&d->h_fin2, 3 * sizeof(double), hipHostMallocDefault));
    MAGMA_CHECK_HIP(hipEventCreateWithFlags(&d->pre_ev, hipEventDisableTiming));
    MAGMA_CHECK_HIP(hipEventCreateWithFlags(&d->pre0_ev, hipEventDisableTiming));
  }
  if (c->has_comm()) {
    // all-reduce staging: lower 64 x 64 tiles of P, contiguous; the gathered theta_i
    if ((rc = d->pack.alloc(sizeof(double) * (size_t)nt * (nt + 1) / 2 * 4096)) ||
        (rc = d->gath.alloc(sizeof(double) * 3 * Mt)))
      return rc;
    // lower tiles (i, j <= i) of the theta_0 gradient owned by this rank: block rows
    // i = rank, rank + world, ... (cyclic: row i has i + 1 tiles)
    std::vector<int> pl;
    for (int i = c->rank; i < nt; i += c->world)
      for (int j = 0; j <= i; ++j) pl.push_back(i * (i + 1) / 2 + j);
    d->nplist = (int)pl.size();
    if ((rc = upload(d->plist, pl))) return rc;
  }
  if (!d->large.empty()) {
    d->np_large = magma_pad(nmaxL);
    const size_t npl = (size_t)d->np_large;
    const std::vector<int64_t> z2 = {0, 0};
    if ((rc = d->LK.alloc(sizeof(double) * npl * npl)) || (rc = d->LKc.alloc(sizeof(double) * npl * npl)) ||
        (rc = d->LC.alloc(sizeof(double) * npl * npl)) || (rc = d->Lm.alloc(sizeof(double) * npl)) ||
        (rc = d->Lwork.alloc(magma_workspace_bytes(nmaxL))) || (rc = d->Lscal.alloc(sizeof(double) * 8)) ||
        (rc = upload(d->zero2, z2)))
      return rc;
    for (auto& L : d->large) {
      const std::vector<int64_t> offL = {0, (int64_t)L->n};
      if ((rc = upload(L->t, L->h_t)) || (rc = upload(L->y, L->h_y)) || (rc = upload(L->idx, L->h_idx)) ||
          (rc = upload(L->off, offL)))
        return rc;
    }
  }
  // (on the context's stream and waited for: hipMemset runs on the null stream, which the library's
  // non-blocking streams do not wait for -- with other threads keeping the null stream busy it
  // could land after the first set_m0 and wipe the prior mean)
  MAGMA_CHECK_HIP(hipMemsetAsync(d->mean.p, 0, sizeof(double) * Np, c->main));
  MAGMA_CHECK_HIP(hipMemsetAsync(d->m0.p, 0, sizeof(double) * Np, c->main));
  if ((rc = sync(c->main))) return rc;
  *out = guard.release();
  return 0;
}

void magma_dataset_destroy(magma_dataset* d) {
  if (!d) return;
  if (d->ctx) (void)hipSetDevice(d->ctx->device);
  delete d;
}

int magma_dataset_info(const magma_dataset* d, int64_t* M, int* nmax, int* N, int* Np) {
  if (!d) return fail(2, "dataset is NULL");
  if (M) *M = d->Mloc;
  if (nmax) {
    *nmax = d->nmax;
    for (auto& L : d->large) *nmax = std::max(*nmax, L->n);
  }
  if (N) *N = d->N;
  if (Np) *Np = d->Np;
  return 0;
}

void* magma_dataset_buffer(const magma_dataset* d, const char* name) {
  if (!d || !name) return nullptr;
  const std::string s(name);
  if (s == "mean") return d->mean.p;
  if (s == "cov") return d->P.p;
  if (s == "m0") return d->m0.p;
  if (s == "grid") return d->grid.p;
  if (s == "work") {  // (the caller may write into it: the cached K_0^-1 is no longer trusted)
    const_cast<magma_dataset*>(d)->a0_valid = false;
    return d->work.p;
  }
  if (s == "w") return d->w.p;
  return nullptr;
}

int magma_get_mean(magma_ctx* c, magma_dataset* d, double* h_mean) {
  if (!c || !d || !h_mean) return fail(2, "get_mean: NULL argument");
  if (int rc = set_device(c)) return rc;
  if (int rc = d2h(h_mean, d->mean.p, sizeof(double) * (size_t)d->N, c->main)) return rc;
  return sync(c->main);
}

int magma_get_cov(magma_ctx* c, magma_dataset* d, double* h_cov, int64_t ld) {
  if (!c || !d || !h_cov) return fail(2, "get_cov: NULL argument");
  if (ld < d->N) return fail(2, "get_cov: ld < N");
  if (int rc = set_device(c)) return rc;
  MAGMA_CHECK_HIP(hipMemcpy2DAsync(h_cov, (size_t)ld * sizeof(double), d->P.p, (size_t)d->Np * sizeof(double),
                                   (size_t)d->N * sizeof(double), (size_t)d->N, hipMemcpyDeviceToHost, c->main));
  return sync(c->main);
}

int magma_set_mean_cov(magma_ctx* c, magma_dataset* d, const double* h_mean, const double* h_cov, int64_t ld) {
  if (!c || !d || !h_mean || !h_cov) return fail(2, "set_mean_cov: NULL argument");
  if (int rc = set_device(c)) return rc;
  hipStream_t st = c->main;
  MAGMA_CHECK_HIP(hipMemsetAsync(d->mean.p, 0, sizeof(double) * (size_t)d->Np, st));
  if (int rc = h2d(d->mean.p, h_mean, sizeof(double) * (size_t)d->N, st)) return rc;
  if (int rc = launch_fill((int64_t)d->Np * d->Np, 0.0, d->P.as<double>(), st)) return rc;
  if (int rc = launch_set_identity_pad(d->N, d->Np, d->P.as<double>(), st)) return rc;
  MAGMA_CHECK_HIP(hipMemcpy2DAsync(d->P.p, (size_t)d->Np * sizeof(double), h_cov, (size_t)ld * sizeof(double),
                                   (size_t)d->N * sizeof(double), (size_t)d->N, hipMemcpyHostToDevice, st));
  d->cpack_ok = false;
  d->dedup_ok = false;
  d->covfac_ok = false;
  return sync(st);
}

int magma_set_m0(magma_ctx* c, magma_dataset* d, const double* h_m0, int n_m0) {
  if (!c || !d || !h_m0) return fail(2, "set_m0: NULL argument");
  if (int rc = set_device(c)) return rc;
  return set_m0(d, h_m0, n_m0, c->main);
}

int magma_get_stats(const magma_dataset* d, magma_em_stats* out, int reset) {
  if (!d || !out) return fail(2, "get_stats: NULL argument");
  if (d->flag_pending) {  // (an EM iteration that ended between its E-step and M-step)
    magma_dataset* dd = const_cast<magma_dataset*>(d);
    if (int rc = count_flagged(dd, dd->ctx->main)) return rc;
  }
  *out = d->stats;
  if (reset) const_cast<magma_dataset*>(d)->stats = magma_em_stats{0, 0, 0, 0, 0, 0};
  return 0;
}

// ------------------------------------------------------------------ E-step (CF:371-395)
int magma_e_step(magma_ctx* c, magma_dataset* d, const double* h_theta0, const double* h_theta_i,
                 int theta_stride, double pen_diag, const double* h_m0, int n_m0, int* h_info,
                 double* h_f0) {
  if (!c || !d || !h_theta0 || !h_theta_i || !h_m0) return fail(2, "e_step: NULL argument");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (int rc = set_device(c)) return rc;
  hipStream_t st = c->main;
  int rc;
  HT("(between calls)");
  if ((rc = set_m0(d, h_m0, n_m0, st))) return rc;
  HT("e: set_m0");
  // the dense part of this very E-step may already be queued (or done): speculate_estep
  const bool hit = d->spec_issued && theta_stride == 3 && d->spec_key[0] == h_theta0[0] &&
                   d->spec_key[1] == h_theta0[1] && d->spec_key[2] == pen_diag && d->spec_m0 == d->h_m0_last &&
                   d->pre_theta.size() == (size_t)(3 * d->M_total) &&
                   std::equal(d->pre_theta.begin(), d->pre_theta.end(), h_theta_i);
  // (on a hit the rows are usually on the device already: theta_dev_valid)
  if (!(hit && d->theta_dev_valid) && (rc = set_theta(d, h_theta_i, theta_stride, st))) return rc;
  HT("e: set_theta");
  if (d->spec_issued) {
    d->spec_issued = false;
    if (hit) {
      MAGMA_CHECK_HIP(hipEventSynchronize(d->spec_ev));
      std::swap(d->P.p, d->P2.p);
      std::swap(d->w.p, d->w2.p);
      std::swap(d->info.p, d->info2.p);
      std::swap(d->mean.p, d->mean2.p);
      int info = 0;
      double f0 = 0.0;
      estep_finish_read(d->h_fin2, &f0, &info);
      HT("e: finish (sync)");
      d->a0_valid = (info == 0);
      d->a0_key[0] = h_theta0[0];
      d->a0_key[1] = h_theta0[1];
      d->a0_key[2] = pen_diag;
      d->last_f0 = f0;
      d->dedup_ok = false;
      d->cpack_ok = (info == 0) && d->spec_cpack;
      d->covfac_ok = (info == 0) && d->spec_fac;
      d->flag_pending = true;
      if ((rc = count_flagged(d, st, false))) return rc;
      if (info == 0 && d->cpack_early && !d->cpack_ok && !resident_gather_enabled() && (rc = ensure_cpack(d, st))) return rc;
      if (h_info) *h_info = info;
      if (h_f0) *h_f0 = f0;
      HT("e: factor + flagged (sync)");
      return 0;
    }
    // not taken: its kernels are ordered on this stream ahead of everything queued below, and the
    // prefetched sums it consumed are gone (pre_valid is already false)
  }
  d->cpack_ok = false;
  d->dedup_ok = false;
  const size_t Np = (size_t)d->Np;
  const bool pre = d->pre_valid && theta_stride == 3 &&
                   std::equal(d->pre_theta.begin(), d->pre_theta.end(), h_theta_i);
  d->pre_valid = false;
  if (d->pre_issued) {  // (also when it is not taken: it writes Kinv / v, which the sums below use)
    MAGMA_CHECK_HIP(hipStreamWaitEvent(st, d->pre_ev, 0));
    d->pre_issued = false;
  }
  if (pre) {
    std::swap(d->P.p, d->P2.p);
    std::swap(d->w.p, d->w2.p);
    std::swap(d->info.p, d->info2.p);
  } else {
    MAGMA_CHECK_HIP(hipMemsetAsync(d->P.p, 0, sizeof(double) * Np * Np, st));
    MAGMA_CHECK_HIP(hipMemsetAsync(d->w.p, 0, sizeof(double) * Np, st));
  }
  if (d->M > 0 && !pre &&
      (rc = magma_estep_gather((int)d->M, d->nmax, d->off.as<int64_t>(), d->off2.as<int64_t>(),
                               d->t.as<double>(), d->y.as<double>(), d->theta.as<double>(), theta_stride,
                               d->Kinv.as<double>(), d->v.as<double>(), d->f.as<double>(), d->nP,
                               d->ptrP.as<int64_t>(), d->srcP.as<int>(), d->tgtP.as<int64_t>(), d->nw,
                               d->ptrw.as<int64_t>(), d->srcw.as<int>(), d->tgtw.as<int64_t>(),
                               d->P.as<double>(), d->w.as<double>(), d->info.as<int>(), st)))
    return rc;
  for (auto& L : d->large) {
    // n_i > 64: K_i and its inverse on the dense kernels (kern_to_inv, CF:141-170), then
    // update_inv / update_mean (CF:440-478) on that explicit block
    const double* th = theta_row(d, h_theta_i, theta_stride, L->row);
    double* sc = d->Lscal.as<double>();
    int* li = reinterpret_cast<int*>(sc + 2);
    if ((rc = magma_kern_to_cov_grid(L->n, L->t.as<double>(), th[0], th[1], th[2], d->LK.as<double>(), st)) ||
        (rc = magma_spd_inverse(L->n, d->LK.as<double>(), sc, li, d->Lwork.p, st)))
      return rc;
    int linfo = 0;
    if ((rc = d2h(&linfo, li, sizeof(int), st)) || (rc = sync(st))) return rc;
    if (linfo != 0) {
      // The solve() / ginv() slow path (CF:161) covers n_i <= 64 only. Leaving this individual out
      // of the hyper-posterior would be a silent deviation from the reference: an error instead.
      d->stats.flagged += 1;
      return fail(5, "E-step: K_i of an individual with more than 64 observations is not positive definite "
                     "(local row " + std::to_string(L->row) + ", info=" + std::to_string(linfo) +
                     "); the pseudo-inverse route exists for n_i <= 64 only");
    }
    if ((rc = magma_copy_unpad(L->n, d->LK.as<double>(), d->LKc.as<double>(), L->n, st)) ||
        (rc = launch_update_inv(1, L->off.as<int64_t>(), d->zero2.as<int64_t>(), L->idx.as<int>(),
                                d->LKc.as<double>(), L->y.as<double>(), d->P.as<double>(), d->Np,
                                d->w.as<double>(), st)))
      return rc;
  }
  if (c->has_comm()) {
    // partial precision sums: only the lower tiles carry anything -- packed, summed, unpacked
    const int nt = d->Np / 64;
    const int64_t npk = (int64_t)nt * (nt + 1) / 2 * 4096;
    if ((rc = launch_pack_lower(d->Np, d->P.as<double>(), d->pack.as<double>(), 0, st))) return rc;
    if ((rc = c->allreduce(d->pack.as<double>(), npk, st))) return rc;
    if ((rc = launch_pack_lower(d->Np, d->P.as<double>(), d->pack.as<double>(), 1, st))) return rc;
    if ((rc = c->allreduce(d->w.as<double>(), d->Np, st))) return rc;
  }
  HT("e: sums launched");
  const int reuse = d->a0_valid && d->a0_key[0] == h_theta0[0] && d->a0_key[1] == h_theta0[1] &&
                    d->a0_key[2] == pen_diag;
  int info = 0;
  double f0 = 0.0;
  if (!d->h_fin) MAGMA_CHECK_HIP(hipHostMalloc((void**)&d->h_fin, 3 * sizeof(double), hipHostMallocDefault));
  if ((rc = estep_finish_launch(d->N, d->grid.as<double>(), h_theta0[0], h_theta0[1], pen_diag,
                                d->m0.as<double>(), d->P.as<double>(), d->w.as<double>(),
                                d->mean.as<double>(), d->work.p, reuse, 1, d->h_fin, st)))
    return rc;
  // Queued behind the dense part BEFORE the host waits for it (a failed factorisation leaves
  // garbage in both, marked invalid below): inside an EM iteration with individual
  // hyper-parameters the M-step's first need is the packed cov blocks; and the factor of cov for
  // the theta_0 gradient.
  const bool want_fac = (!c->has_comm() || !theta0_shard_enabled()) && d->covfac.p && covfac_enabled();
  if (d->cpack_early && !resident_gather_enabled() && (rc = ensure_cpack(d, st))) return rc;
  if (want_fac && (rc = estep_cov_factor(d->N, d->work.p, d->covfac.as<double>(), st))) return rc;
  if (d->M > 0) {  // individuals whose K_i was not positive definite (they took the solve() / ginv() slow path)
    if (!d->h_flag)
      MAGMA_CHECK_HIP(hipHostMalloc((void**)&d->h_flag, sizeof(int) * (size_t)d->M, hipHostMallocDefault));
    MAGMA_CHECK_HIP(hipMemcpyAsync(d->h_flag, d->info.p, sizeof(int) * (size_t)d->M, hipMemcpyDeviceToHost, st));
    d->flag_pending = true;
  }
  if ((rc = sync(st))) return rc;
  estep_finish_read(d->h_fin, &f0, &info);
  HT("e: finish (sync)");
  d->a0_valid = (info == 0);
  d->a0_key[0] = h_theta0[0];
  d->a0_key[1] = h_theta0[1];
  d->a0_key[2] = pen_diag;
  d->last_f0 = f0;
  if (info != 0) d->cpack_ok = false;
  d->covfac_ok = (info == 0) && want_fac;
  if (d->M > 0 && (rc = count_flagged(d, st, false))) return rc;  // (copied before the wait above)
  if (h_info) *h_info = info;
  if (h_f0) *h_f0 = f0;
  HT("e: factor + flagged (sync)");
  return 0;
}

// ------------------------------------------------------------------ logL_monitoring (CF:236-264)
int magma_logl_monitoring(magma_ctx* c, magma_dataset* d, const double* h_theta0,
                          const double* h_theta_i, int theta_stride, double* h_out) {
  if (!c || !d || !h_theta0 || !h_theta_i || !h_out) return fail(2, "logl_monitoring: NULL argument");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (int rc = set_device(c)) return rc;
  const double pen = mean_sigma(d, h_theta_i, theta_stride, 1.0);  // CF:251
  double ll0 = 0.0, s4[4];
  if (int rc = theta0_f(c, d, h_theta0, pen, &ll0)) return rc;
  // equal rows = common hyper-parameters (CF:245-249 then takes the *_common_hp sum)
  bool same = theta_stride == 3 && d->common_times;
  for (int64_t i = 1; i < d->M_total && same; ++i)
    same = std::equal(h_theta_i, h_theta_i + 3, h_theta_i + 3 * i);
  if (int rc = sum_fi(c, d, h_theta_i, same ? 0 : theta_stride, 0, s4)) return rc;
  *h_out = -ll0 - s4[0];
  return 0;
}

// ------------------------------------------------------------------ sums over individuals
int magma_sum_fi(magma_ctx* c, magma_dataset* d, const double* h_theta_i, int theta_stride,
                 int want_grad, double* h_out4) {
  if (!c || !d || !h_theta_i || !h_out4) return fail(2, "sum_fi: NULL argument");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (int rc = set_device(c)) return rc;
  return sum_fi(c, d, h_theta_i, theta_stride, want_grad ? 1 : 0, h_out4);
}

// ------------------------------------------------------------------ M-step (CF:397-437)
int magma_m_step(magma_ctx* c, magma_dataset* d, const double* h_theta0, const double* h_theta_i,
                 int common_hp, const double* opt4, double* h_theta0_out, double* h_theta_i_out) {
  if (!c || !d || !h_theta0 || !h_theta_i || !h_theta0_out || !h_theta_i_out)
    return fail(2, "m_step: NULL argument");
  if (int rc = set_device(c)) return rc;
  const Opt o(opt4);
  const double pen = mean_sigma(d, h_theta_i, 3, 2.0);  // CF:409: mean(2 sigma_i)
  hipStream_t st = c->main;
  const int64_t M = d->M, Mt = d->M_total;

  // theta_0 (CF:412-413): a few dense N x N evaluations, each a chain of small dependent
  // kernels -- on the high-priority side stream in a helper thread, next to the theta_i problems
  // (independent given the E-step). With several ranks the two N^3 products of its gradient are
  // split by block rows and their two sums all-reduced per evaluation.
  int rc0 = 0;
  std::string err0;
  double th0_new[2] = {h_theta0[0], h_theta0[1]};
  int nfev0 = 0;
  // the first evaluation's K_0^-1 may already be in the workspace (magma_em_iteration)
  const bool pre0 = d->pre0_issued && d->pre0_key[0] == h_theta0[0] && d->pre0_key[1] == h_theta0[1] &&
                    d->pre0_key[2] == pen;
  if (d->pre0_issued) {  // (taken or not: it writes the workspace the chain is about to use)
    MAGMA_CHECK_HIP(hipStreamWaitEvent(c->side, d->pre0_ev, 0));
    d->pre0_issued = false;
  }
  auto run0 = [&]() {
    (void)hipSetDevice(c->device);
    Theta0Shard sh{c->rank, c->world, d->plist.as<int>(), d->nplist,
                   [](void* u, double* p, int64_t n, hipStream_t s) {
                     return static_cast<magma_ctx*>(u)->allreduce(p, n, s);
                   }, c};
    rc0 = theta0_mstep_impl(d->N, d->grid.as<double>(), h_theta0, pen, d->mean.as<double>(),
                            d->m0.as<double>(), d->P.as<double>(), o.m, o.factr, o.pgtol, o.maxit,
                            d->work0.p, th0_new, nullptr, nullptr, nullptr, &nfev0, nullptr, c->side,
                            (c->has_comm() && theta0_shard_enabled()) ? &sh : nullptr,
                            d->covfac_ok ? d->covfac.as<double>() : nullptr, pre0);
    if (rc0) err0 = magma_last_error();
  };
  // theta_0 and the theta_i problems overlap only when neither side needs the communicator at
  // the same time: individual hyper-parameters (the theta_i problems exchange nothing until the
  // end), or a single rank
  HT("m: entry");
  const bool overlap = !common_hp || !c->has_comm();
  std::thread th0;
  bool started0 = false;
  auto start0 = [&]() {  // (idempotent)
    if (overlap && !started0) {
      started0 = true;
      th0 = std::thread(run0);
    }
  };
  if (overlap) {
    // both streams read the hyper-posterior the E-step left on the main stream
    if (int rc = sync(st)) return rc;
    if (int rc = count_flagged(d, st, false)) return rc;
    const bool split = !common_hp && c->reserve_cus > 0 && M > 0;
    set_resident_max_workgroups_thread(split ? std::max(1, c->ncu - c->reserve_cus) : -1);
    // with individual hyper-parameters the resident kernel is launched first (a few microseconds)
    // and the helper thread -- tens of microseconds to create -- right behind it
    if (common_hp || M == 0) start0();
  }
  bool joined = false;
  auto join0 = [&]() -> int {
    if (th0.joinable()) th0.join();
    if (!joined) {
      joined = true;
      set_resident_max_workgroups_thread(-1);
      d->stats.theta0_evals += nfev0;
    }
    if (rc0) return fail(rc0, err0);
    return 0;
  };
  int rc = 0;
  double sums[2] = {0.0, 0.0};
  if (common_hp) {
    // CF:415-420: one 3-parameter problem, objective = sum over ALL individuals
    magma_lbfgsb* lb = magma_lbfgsb_create(1, 3, h_theta_i, o.m, o.factr, o.pgtol, o.maxit);
    if (!lb) rc = fail(2, "m_step: bad optimiser parameters");
    bool first = true;
    while (!rc) {
      double x[3];
      unsigned char act = 0;
      int na = 0;
      if ((rc = magma_lbfgsb_ask(lb, x, &act, &na)) || na == 0) break;
      double s4[4];
      if ((rc = sum_fi(c, d, x, 0, 1, s4))) break;
      d->stats.common_evals += 1;
      if (first) sums[0] = s4[0];
      first = false;
      if ((rc = magma_lbfgsb_tell(lb, s4, s4 + 1))) break;
    }
    double par[3] = {0, 0, 0}, fbest = 0.0;
    if (!rc) rc = magma_lbfgsb_result(lb, par, &fbest, nullptr, nullptr, nullptr);
    if (lb) magma_lbfgsb_destroy(lb);
    sums[1] = fbest;
    if (!rc)
      for (int64_t i = 0; i < Mt; ++i)
        for (int q = 0; q < 3; ++q) h_theta_i_out[3 * i + q] = par[q];
    if (!overlap && !rc) run0();
  } else {
    std::vector<double> hx((size_t)(3 * std::max<int64_t>(M, 1))), hf((size_t)std::max<int64_t>(M, 1)),
        hfx0((size_t)std::max<int64_t>(M, 1));
    std::vector<int> hnf((size_t)std::max<int64_t>(M, 1));
    if (M > 0) {
      do {
        HT("m: sync + thread start");
        // the packed cov blocks: gathered by the resident kernel itself, each block by the
        // wavefront that starts that individual (no launch between the E-step and this kernel)
        const bool in_kernel_gather = !d->cpack_ok && resident_gather_enabled();
        if (!in_kernel_gather && (rc = ensure_cpack(d, st))) break;
        if (!d->theta_from_estep && (rc = set_theta(d, h_theta_i, 3, st))) break;
        if (in_kernel_gather) set_resident_gather_thread(d->P.as<double>(), d->Np);
        HT("m: cpack + theta");
        const int* order = nullptr;
        // longest chains of the previous call first (they tend to stay the longest); the
        // permutation was uploaded at the end of that call
        if (d->have_order) order = d->order.as<int>();
        HT("m: order");
        if ((rc = magma_indiv_mstep_resident(0, (int)M, d->nmax, d->off.as<int64_t>(), d->t.as<double>(),
                                             d->y.as<double>(), d->idx.as<int>(), d->mean.as<double>(),
                                             d->cpack.as<double>(), d->theta.as<double>(), order, o.m,
                                             o.factr, o.pgtol, o.maxit, d->rres.as<double>(),
                                             d->rres.as<double>() + 3 * M, d->rres.as<double>() + 4 * M,
                                             d->rit.as<int>(),
                                             reinterpret_cast<int*>(d->rres.as<double>() + 5 * M),
                                             reinterpret_cast<int*>(d->rres.as<double>() + 5 * M) + M, st)))
          break;
        if (in_kernel_gather) d->cpack_ok = true;  // (valid once this stream has passed the kernel)
        start0();
        d->m_redo = false;
        d->pre_early = false;
        if (d->want_early && (rc = early_estep_sums(c, d))) break;
        // x, f, f(x0) and the evaluation counts sit in one block: one copy into pinned memory
        if (!d->h_rres &&
            (rc = check_hip(hipHostMalloc(&d->h_rres, sizeof(double) * 6 * (size_t)M, hipHostMallocDefault),
                            "hipHostMalloc")))
          break;
        if ((rc = d2h(d->h_rres, d->rres.p, sizeof(double) * 6 * (size_t)M, st)) || (rc = sync(st))) break;
        {
          const double* hr = static_cast<const double*>(d->h_rres);
          std::copy(hr, hr + 3 * M, hx.begin());
          std::copy(hr + 3 * M, hr + 4 * M, hf.begin());
          std::copy(hr + 4 * M, hr + 5 * M, hfx0.begin());
          const int* hn = reinterpret_cast<const int*>(hr + 5 * M);
          std::copy(hn, hn + M, hnf.begin());
          // status 6: one of the problem's covariances was not positive definite for the sweep.
          // The reference evaluates those through solve() / ginv() (CF:133): such problems are
          // solved again, from their start, by the lock-step route, whose evaluations do that.
          std::vector<int> redo;
          for (int64_t i = 0; i < M; ++i)
            if (hn[M + i] == 6) redo.push_back((int)i);
          if (!redo.empty()) {
            d->m_redo = true;
            const int nr = (int)redo.size();
            std::vector<double> x0((size_t)3 * nr), xr((size_t)3 * nr), fr((size_t)nr), f0r((size_t)nr);
            std::vector<int> nfr((size_t)nr);
            for (int k = 0; k < nr; ++k) {
              const double* row = theta_row(d, h_theta_i, 3, d->small_row[(size_t)redo[(size_t)k]]);
              std::copy(row, row + 3, x0.begin() + 3 * k);
            }
            if ((rc = d->redo_theta.ensure(sizeof(double) * 3 * (size_t)nr))) break;
            if ((rc = indiv_mstep_lockstep(0, (int)M, d->nmax, d->off.as<int64_t>(), d->t.as<double>(),
                                           d->y.as<double>(), d->idx.as<int>(), d->mean.as<double>(),
                                           d->cpack.as<double>(), x0.data(), redo.data(), nr, o.m, o.factr,
                                           o.pgtol, o.maxit, d->redo_theta.as<double>(), d->sel.as<int>(),
                                           d->f.as<double>(), d->g.as<double>(), d->info.as<int>(),
                                           xr.data(), fr.data(), f0r.data(), nullptr, nfr.data(), nullptr,
                                           nullptr, nullptr, st)))
              break;
            for (int k = 0; k < nr; ++k) {
              const size_t i = (size_t)redo[(size_t)k];
              std::copy(xr.begin() + 3 * k, xr.begin() + 3 * k + 3, hx.begin() + 3 * i);
              hf[i] = fr[(size_t)k];
              hfx0[i] = f0r[(size_t)k];
              hnf[i] += nfr[(size_t)k];
            }
            d->stats.slow_resolved += nr;
          }
        }
        HT("m: resident (launch + d2h sync)");
        {  // next call's order: by evaluation count, descending, stable (counting sort)
          int top = 0;
          for (int64_t i = 0; i < M; ++i) top = std::max(top, hnf[(size_t)i]);
          std::vector<int> pos((size_t)top + 2, 0);
          for (int64_t i = 0; i < M; ++i) ++pos[(size_t)(top - hnf[(size_t)i]) + 1];
          for (size_t q = 1; q < pos.size(); ++q) pos[q] += pos[q - 1];
          d->h_perm.resize((size_t)M);
          for (int64_t i = 0; i < M; ++i) d->h_perm[(size_t)pos[(size_t)(top - hnf[(size_t)i])]++] = (int)i;
          if ((rc = h2d(d->order.p, d->h_perm.data(), sizeof(int) * (size_t)M, st))) break;
          d->have_order = true;
        }
        int mx = 0;
        for (int64_t i = 0; i < M; ++i) {
          sums[0] += hfx0[(size_t)i];
          sums[1] += hf[(size_t)i];
          d->stats.indiv_evals += hnf[(size_t)i];
          mx = std::max(mx, hnf[(size_t)i]);
        }
        d->stats.indiv_rounds += mx;
      } while (false);
    }
    start0();  // (M == 0, or an error before the launch: the theta_0 problem is still solved)
    // results by local row; individuals on the dense route are optimised one after the other by
    // the host-side L-BFGS-B around the dense evaluation (CF:423-431 for n_i > 64)
    std::vector<double> loc((size_t)(3 * std::max<int64_t>(d->Mloc, 1)));
    for (int64_t k = 0; k < M; ++k)
      for (int q = 0; q < 3; ++q) loc[(size_t)(3 * d->small_row[(size_t)k] + q)] = hx[(size_t)(3 * k + q)];
    for (auto& L : d->large) {
      if (rc) break;
      const double* start = theta_row(d, h_theta_i, 3, L->row);
      magma_lbfgsb* lb = magma_lbfgsb_create(1, 3, start, o.m, o.factr, o.pgtol, o.maxit);
      if (!lb) {
        rc = fail(2, "m_step: bad optimiser parameters");
        break;
      }
      bool first = true;
      while (!rc) {
        double x[3], o4[4];
        unsigned char act = 0;
        int na = 0, info = 0;
        if ((rc = magma_lbfgsb_ask(lb, x, &act, &na)) || na == 0) break;
        if ((rc = large_eval(c, d, *L, x, 1, o4, &info))) break;
        if (first) sums[0] += o4[0];
        first = false;
        if ((rc = magma_lbfgsb_tell(lb, o4, o4 + 1))) break;
      }
      double fb = 0.0;
      if (!rc) rc = magma_lbfgsb_result(lb, &loc[(size_t)(3 * L->row)], &fb, nullptr, nullptr, nullptr);
      magma_lbfgsb_destroy(lb);
      sums[1] += fb;
    }
    HT("m: host sums");
    // the theta_0 thread may still be using the communicator: wait for it before this thread does
    if (int r2 = join0()) rc = rc ? rc : r2;
    HT("m: join theta_0");
    if (g_ht.on && getenv("MAGMA_HOST_TIMING")[0] == '2')
      fprintf(stderr, "[magma m-step] theta_0 evaluations %d, wait for the theta_0 chain after the resident kernel %.2f ms\n",
              nfev0, g_ht.acc["m: join theta_0"].second * 1e-3 - g_ht_prev_join);
    if (g_ht.on) g_ht_prev_join = g_ht.acc["m: join theta_0"].second * 1e-3;
    if (!rc && c->has_comm()) {
      double* gb = d->gath.as<double>();
      rc = check_hip(hipMemsetAsync(gb, 0, sizeof(double) * 3 * (size_t)Mt, st), "m_step gather");
      if (!rc && d->Mloc > 0) rc = h2d(gb + 3 * d->row0, loc.data(), sizeof(double) * 3 * (size_t)d->Mloc, st);
      if (!rc) rc = c->allreduce(gb, 3 * Mt, st);
      if (!rc) rc = d2h(h_theta_i_out, gb, sizeof(double) * 3 * (size_t)Mt, st);
      double* fs = d->fsum.as<double>();
      if (!rc) rc = h2d(fs, sums, 2 * sizeof(double), st);
      if (!rc) rc = c->allreduce(fs, 2, st);
      if (!rc) rc = d2h(sums, fs, 2 * sizeof(double), st);
      if (!rc) rc = sync(st);
    } else if (!rc) {
      std::copy(loc.begin(), loc.begin() + 3 * d->Mloc, h_theta_i_out);
    }
  }
  if (int r2 = join0()) rc = rc ? rc : r2;
  if (rc) return rc;
  d->last_sum_fi = sums[0];
  d->last_sum_fi_new = sums[1];
  h_theta0_out[0] = th0_new[0];
  h_theta0_out[1] = th0_new[1];
  HT("m: tail");
  return 0;
}

// K_0^-1 of the M-step's first theta_0 evaluation (CF:409: pen_diag = mean(2 sigma_i)) depends on
// nothing the E-step computes: started on the low-priority stream with a part of the chip, it is
// ready when the theta_0 chain begins. after_monitoring: queued by the previous EM iteration for
// the next one; the low stream then waits for that iteration's monitoring evaluation, so that the
// inverse runs next to the (speculative) E-step's, not next to the monitoring call's.
static int issue_prefactor(magma_ctx* c, magma_dataset* d, const double* h_theta0, const double* h_theta_i,
                           bool after_monitoring) {
  if (!(prefactor_enabled() && d->pre0_ev && d->Np / 64 <= 64 && !d->pre0_issued)) return 0;
  if (int rc = set_device(c)) return rc;
  const double pen_m = mean_sigma(d, h_theta_i, 3, 2.0);
  if (after_monitoring && d->mon_ev) MAGMA_CHECK_HIP(hipStreamWaitEvent(c->low, d->mon_ev, 0));
  if (int rc = theta0_prefactor_impl(d->N, d->grid.as<double>(), h_theta0[0], h_theta0[1], pen_m,
                                     d->work0.p, c->prefactor_wgs, c->low))
    return rc;
  MAGMA_CHECK_HIP(hipEventRecord(d->pre0_ev, c->low));
  d->pre0_issued = true;
  d->pre0_key[0] = h_theta0[0];
  d->pre0_key[1] = h_theta0[1];
  d->pre0_key[2] = pen_m;
  return 0;
}

// ------------------------------------------------------------------ one EM iteration (MG:32-75)
int magma_em_iteration(magma_ctx* c, magma_dataset* d, const double* h_theta0, const double* h_theta_i,
                       const double* h_m0, int n_m0, int common_hp, const double* opt4,
                       double* h_theta0_out, double* h_theta_i_out, double* h_logl2) {
  if (!c || !d || !h_logl2) return fail(2, "em_iteration: NULL argument");
  const double nanv = std::numeric_limits<double>::quiet_NaN();
  const double pen_e = mean_sigma(d, h_theta_i, 3, 1.0);  // CF:381 (signed mean)
  int info = 0;
  if (int rc = issue_prefactor(c, d, h_theta0, h_theta_i, false)) return rc;
  d->cpack_early = !common_hp;
  const int rc_e = magma_e_step(c, d, h_theta0, h_theta_i, 3, pen_e, h_m0, n_m0, &info, nullptr);
  d->cpack_early = false;
  if (rc_e) return rc_e;
  d->theta_from_estep = !common_hp && d->large.empty();  // (same h_theta_i: the M-step need not upload it again)
  if (info != 0) {
    std::string extra;
    if (info == 999 || info == -999) {  // the tile-graph inverse gave up on a flag: say which
      int dg[2] = {0, 0};
      (void)dag_diag(d->N, d->work.p, dg);
      extra = "; tile graph abandoned: late steps " + std::to_string(dg[0]) + ", flag index " +
              std::to_string(dg[1] & 0xffff) + ", workgroup " + std::to_string(dg[1] >> 16) +
              ", nt " + std::to_string(d->Np / 64);
    }
    return fail(5, std::string("E-step: ") + (info > 0 ? "K_0" : "hyper-posterior precision") +
                       " not positive definite (info=" + std::to_string(info) + ")" + extra);
  }
  d->want_early = !common_hp;  // the next E-step's sums right behind the resident kernel
  const int rc_m = magma_m_step(c, d, h_theta0, h_theta_i, common_hp, opt4, h_theta0_out, h_theta_i_out);
  d->want_early = false;
  d->theta_from_estep = false;
  if (rc_m) return rc_m;
  bool bad = std::isnan(h_theta0_out[0]) || std::isnan(h_theta0_out[1]);
  for (int64_t k = 0; k < 3 * d->M_total && !bad; ++k) bad = std::isnan(h_theta_i_out[k]);
  if (bad) {  // MG:46-51
    h_logl2[0] = h_logl2[1] = nanv;
    return 0;
  }
  // logL_monitoring(hp) (MG:54) from by-products: its theta_0 term came out of the E-step's
  // K_0^-1 (same theta_0, same pen_diag: CF:251, 381), its individual terms are the optimisers'
  // first evaluations. logL_monitoring(new hp) (MG:60): individual terms = the optimisers' final
  // values; only the theta_0 term needs an evaluation, which leaves K_0^-1 for the next E-step.
  h_logl2[0] = -d->last_f0 - d->last_sum_fi;
  HT("em: nan scan");
  // the monitoring evaluation is queued first and waited for last: the host work of the next
  // E-step's sums (queued on the side stream) runs under it
  const double sig = mean_sigma(d, h_theta_i_out, 3, 1.0);
  if (!d->h_mon) MAGMA_CHECK_HIP(hipHostMalloc((void**)&d->h_mon, 4 * sizeof(double), hipHostMallocDefault));
  // (a capped grid leaves room for the next E-step's sums when they are launched next to this
  // inverse; sums that were started behind the resident kernel are mostly over by now)
  const bool will_prefetch = d->M > 0 && prefetch_enabled() && !(d->pre_early && d->pre_issued && !d->m_redo);
  if (will_prefetch && c->monitor_wgs > 0) set_dag_grid_once(c->monitor_wgs);
  d->stats.theta0_evals += 1;
  d->a0_valid = false;
  const int rc_mon = theta0_eval_impl(d->N, d->grid.as<double>(), h_theta0_out[0], h_theta0_out[1], sig,
                                      d->mean.as<double>(), d->m0.as<double>(), d->P.as<double>(), 0,
                                      d->work.p, nullptr, nullptr, c->main, nullptr, nullptr, false, d->h_mon);
  set_dag_grid_once(0);
  if (rc_mon) return rc_mon;
  if (d->mon_ev) MAGMA_CHECK_HIP(hipEventRecord(d->mon_ev, c->main));
  HT("em: monitoring launch");
  // the next E-step's sums: already running if they were started behind the resident kernel and
  // its theta_i rows were the final ones (no problem solved again on the host)
  const bool early_ok = d->pre_early && d->pre_issued && !d->m_redo;
  d->pre_early = false;
  if (early_ok) {
    d->pre_theta.assign(h_theta_i_out, h_theta_i_out + 3 * d->M_total);
    d->pre_valid = true;
    if (d->large.empty() && !c->has_comm()) {
      // the rows the next E-step / M-step start from are the resident kernel's results as they
      // sit in device memory (h_theta_i_out is a copy of them): no upload on the next call
      MAGMA_CHECK_HIP(hipMemcpyAsync(d->theta.p, d->rres.p, sizeof(double) * 3 * (size_t)d->M,
                                     hipMemcpyDeviceToDevice, c->main));
      d->theta_dev_valid = true;
    }
  } else if (int rc = prefetch_estep_sums(c, d, h_theta_i_out)) {
    return rc;
  }
  HT("em: prefetch launch");
  // ... and its dense part, behind the monitoring evaluation whose K_0^-1 it shares; then the next
  // M-step's first K_0^-1 next to it
  if (int rc = speculate_estep(c, d, h_theta0_out, sig, !common_hp && !resident_gather_enabled())) return rc;
  if (d->spec_issued)
    if (int rc = issue_prefactor(c, d, h_theta0_out, h_theta_i_out, true)) return rc;
  if (d->mon_ev) {
    MAGMA_CHECK_HIP(hipEventSynchronize(d->mon_ev));
  } else if (int rc = sync(c->main)) {
    return rc;
  }
  const int info_mon = (int)d->h_mon[3];
  d->a0_valid = (info_mon == 0);
  d->a0_key[0] = h_theta0_out[0];
  d->a0_key[1] = h_theta0_out[1];
  d->a0_key[2] = sig;
  const double ll0 = info_mon == 0 ? d->h_mon[0] : nanv;
  h_logl2[1] = -ll0 - d->last_sum_fi_new;
  HT("em: monitoring (sync)");
  return 0;
}

// ------------------------------------------------------------------ training (MG:15-82)
int magma_training(magma_ctx* c, magma_dataset* d, const double* h_m0, int n_m0,
                   const double* ini_theta0, const double* ini_theta_i3, int common_hp,
                   int n_loop_max, const double* opt4, double* h_theta0_out, double* h_theta_i_out,
                   int* h_converged, int* h_iterations, double* h_trace3) {
  if (!c || !d || !h_m0 || !ini_theta0 || !ini_theta_i3 || !h_theta0_out || !h_theta_i_out)
    return fail(2, "training: NULL argument");
  const int64_t Mt = d->M_total;
  std::vector<double> thi((size_t)(3 * Mt)), nthi((size_t)(3 * Mt));
  for (int64_t i = 0; i < Mt; ++i)
    for (int q = 0; q < 3; ++q) thi[(size_t)(3 * i + q)] = ini_theta_i3[q];  // MG:26-27
  double th0[2] = {ini_theta0[0], ini_theta0[1]}, nth0[2];
  int cv = 0, it = 0;
  for (it = 1; it <= n_loop_max; ++it) {  // MG:32
    double ll[2];
    if (int rc = magma_em_iteration(c, d, th0, thi.data(), h_m0, n_m0, common_hp, opt4, nth0,
                                    nthi.data(), ll))
      return rc;
    if (std::isnan(ll[0])) break;  // MG:46-51: keep the last valid iteration
    const double eps = (ll[1] - ll[0]) / std::fabs(ll[1]);  // MG:64
    if (h_trace3) {
      h_trace3[3 * (it - 1) + 0] = ll[0];
      h_trace3[3 * (it - 1) + 1] = ll[1];
      h_trace3[3 * (it - 1) + 2] = eps;
    }
    if (eps < 1e-3) {  // MG:66-71
      if (eps > 0) {
        th0[0] = nth0[0];
        th0[1] = nth0[1];
        thi.swap(nthi);
      }
      cv = 1;
      break;
    }
    th0[0] = nth0[0];
    th0[1] = nth0[1];
    thi.swap(nthi);
  }
  h_theta0_out[0] = th0[0];
  h_theta0_out[1] = th0[1];
  std::copy(thi.begin(), thi.end(), h_theta_i_out);
  if (h_converged) *h_converged = cv;
  if (h_iterations) *h_iterations = std::min(it, n_loop_max);
  return 0;
}

// ------------------------------------------------------------------ posterior_mu (MG:107-133)
int magma_posterior_mu(magma_ctx* c, magma_dataset** out, int64_t M, int64_t M_total, int64_t row0,
                       const int64_t* h_off, const double* h_t, const double* h_y, int64_t n_extra,
                       const double* h_extra_t, const double* h_theta0, const double* h_theta_i,
                       int theta_stride, double pen_diag, const double* h_m0_scalar, int* h_N,
                       int* h_info) {
  if (!c || !out || !h_theta0 || !h_theta_i || !h_m0_scalar) return fail(2, "posterior_mu: NULL argument");
  // t_pred = sort(unique(timestamps U db U new_db)) (MG:114): the union over the WHOLE db. With
  // several ranks each holds a shard of db, so the ranks' sorted unique timestamps are exchanged
  // first (an all-gather through the all-reduce: every rank writes its block of a zeroed buffer)
  // and every rank builds the same grid; the sizes of the collectives that follow depend on it.
  std::vector<double> grid;
  const int64_t R = M > 0 ? h_off[M] : 0;
  grid.reserve((size_t)(R + n_extra));
  grid.insert(grid.end(), h_t, h_t + R);
  if (n_extra > 0) grid.insert(grid.end(), h_extra_t, h_extra_t + n_extra);
  std::sort(grid.begin(), grid.end());
  grid.erase(std::unique(grid.begin(), grid.end()), grid.end());
  if (c->has_comm() && c->world > 1) {
    if (int rc = set_device(c)) return rc;
    hipStream_t st = c->main;
    DevBuf cnt;
    std::vector<double> hc((size_t)c->world, 0.0);
    hc[(size_t)c->rank] = (double)grid.size();
    if (int rc = upload(cnt, hc)) return rc;
    if (int rc = c->allreduce(cnt.as<double>(), c->world, st)) return rc;
    if (int rc = d2h(hc.data(), cnt.p, sizeof(double) * hc.size(), st)) return rc;
    if (int rc = sync(st)) return rc;
    int64_t tot = 0, mine = 0;
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) mine = tot;
      tot += (int64_t)hc[(size_t)r];
    }
    std::vector<double> all((size_t)std::max<int64_t>(tot, 1), 0.0);
    std::copy(grid.begin(), grid.end(), all.begin() + mine);
    DevBuf ga;
    if (int rc = upload(ga, all)) return rc;
    if (int rc = c->allreduce(ga.as<double>(), tot, st)) return rc;
    if (int rc = d2h(all.data(), ga.p, sizeof(double) * (size_t)tot, st)) return rc;
    if (int rc = sync(st)) return rc;
    all.resize((size_t)tot);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    grid.swap(all);
    // every rank must now hold the same grid: N and a checksum, checked before any P / w collective
    // (sum over ranks of x and of x^2 equal world x and world x^2 only if all x are equal)
    double chk = 0.0;
    for (double g : grid) chk += g;
    std::vector<double> q = {(double)grid.size(), (double)grid.size() * (double)grid.size(), chk, chk * chk};
    DevBuf dq;
    if (int rc = upload(dq, q)) return rc;
    if (int rc = c->allreduce(dq.as<double>(), 4, st)) return rc;
    std::vector<double> qs(4);
    if (int rc = d2h(qs.data(), dq.p, sizeof(double) * 4, st)) return rc;
    if (int rc = sync(st)) return rc;
    const double w = (double)c->world;
    auto near = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * std::max(1.0, std::fabs(b)); };
    if (!near(qs[0], w * q[0]) || !near(qs[1], w * q[1]) || !near(qs[2], w * q[2]) || !near(qs[3], w * q[3]))
      return fail(2, "posterior_mu: the ranks disagree on the prediction grid (different n_extra / h_extra_t?)");
  }
  if (grid.empty()) return fail(2, "posterior_mu: no timestamps");
  magma_dataset* d = nullptr;
  if (int rc = magma_dataset_create(c, &d, M, M_total, row0, h_off, h_t, h_y, (int)grid.size(), grid.data()))
    return rc;
  int info = 0;
  if (int rc = magma_e_step(c, d, h_theta0, h_theta_i, theta_stride, pen_diag, h_m0_scalar, 1, &info, nullptr)) {
    magma_dataset_destroy(d);
    return rc;
  }
  if (h_N) *h_N = (int)grid.size();
  if (h_info) *h_info = info;
  *out = d;
  return 0;
}

int magma_dataset_grid(const magma_dataset* d, double* h_grid) {
  if (!d || !h_grid) return fail(2, "dataset_grid: NULL argument");
  std::copy(d->h_grid.begin(), d->h_grid.begin() + d->N, h_grid);
  return 0;
}

// ------------------------------------------------------------------ pred_gp against a resident
// hyper-posterior (MG:135-181 for B new individuals; host arrays in, host arrays out)
int magma_pred_gp_resident(magma_ctx* c, magma_dataset* d, int B, const int64_t* h_off_n,
                           const double* h_tn, const double* h_yn, const int64_t* h_off_t,
                           const double* h_ts, const double* h_theta, int theta_stride,
                           double* h_mean_out, double* h_var_out, int* h_info) {
  if (!c || !d || !h_off_n || !h_off_t || !h_theta) return fail(2, "pred_gp_resident: NULL argument");
  if (theta_stride != 0 && theta_stride != 3) return fail(2, "theta_stride must be 0 or 3");
  if (B <= 0) return 0;
  if (int rc = set_device(c)) return rc;
  const int64_t Rn = h_off_n[B], Rt = h_off_t[B];
  int nmax = 1, tmax = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t n = h_off_n[b + 1] - h_off_n[b], T = h_off_t[b + 1] - h_off_t[b];
    if (n < 1 || n > MAGMA_NSMALL_MAX) return fail(2, "pred_gp_resident: 1 <= n_b <= 64 observations per individual");
    nmax = std::max(nmax, (int)n);
    tmax = std::max(tmax, (int)T);
  }
  std::vector<int> in((size_t)Rn), it((size_t)std::max<int64_t>(Rt, 1));
  const double* g0 = d->h_grid.data();
  auto on_grid = [&](double x, int& pos) {
    const double* p = std::lower_bound(g0, g0 + d->N, x);
    if (p == g0 + d->N || *p != x) return false;
    pos = (int)(p - g0);
    return true;
  };
  for (int64_t k = 0; k < Rn; ++k)
    if (!on_grid(h_tn[k], in[(size_t)k])) return fail(6, "pred_gp_resident: subscript out of bounds (an observation timestamp is not on the grid)");
  for (int64_t k = 0; k < Rt; ++k)
    if (!on_grid(h_ts[k], it[(size_t)k])) return fail(6, "pred_gp_resident: subscript out of bounds (a target timestamp is not on the grid)");
  DevBuf on, tn, yn, dn, ot, ts, dt, th, om, ov, di;
  const size_t nth = theta_stride ? (size_t)3 * B : 3;
  int rc;
  if ((rc = on.alloc(sizeof(int64_t) * (B + 1))) || (rc = tn.alloc(sizeof(double) * Rn)) ||
      (rc = yn.alloc(sizeof(double) * Rn)) || (rc = dn.alloc(sizeof(int) * Rn)) ||
      (rc = ot.alloc(sizeof(int64_t) * (B + 1))) || (rc = ts.alloc(sizeof(double) * std::max<int64_t>(Rt, 1))) ||
      (rc = dt.alloc(sizeof(int) * std::max<int64_t>(Rt, 1))) || (rc = th.alloc(sizeof(double) * nth)) ||
      (rc = om.alloc(sizeof(double) * std::max<int64_t>(Rt, 1))) ||
      (rc = ov.alloc(sizeof(double) * std::max<int64_t>(Rt, 1))) || (rc = di.alloc(sizeof(int) * B)))
    return rc;
  hipStream_t st = c->main;
  if ((rc = h2d(on.p, h_off_n, sizeof(int64_t) * (B + 1), st)) || (rc = h2d(tn.p, h_tn, sizeof(double) * Rn, st)) ||
      (rc = h2d(yn.p, h_yn, sizeof(double) * Rn, st)) || (rc = h2d(dn.p, in.data(), sizeof(int) * Rn, st)) ||
      (rc = h2d(ot.p, h_off_t, sizeof(int64_t) * (B + 1), st)) ||
      (Rt > 0 && (rc = h2d(ts.p, h_ts, sizeof(double) * Rt, st))) ||
      (Rt > 0 && (rc = h2d(dt.p, it.data(), sizeof(int) * Rt, st))) ||
      (rc = h2d(th.p, h_theta, sizeof(double) * nth, st)))
    return rc;
  MAGMA_CHECK_HIP(hipMemsetAsync(di.p, 0, sizeof(int) * B, st));
  if ((rc = magma_pred_gp_batched(B, nmax, tmax, on.as<int64_t>(), tn.as<double>(), yn.as<double>(),
                                  dn.as<int>(), ot.as<int64_t>(), ts.as<double>(), dt.as<int>(),
                                  d->mean.as<double>(), d->P.as<double>(), d->N, th.as<double>(),
                                  theta_stride, om.as<double>(), ov.as<double>(), di.as<int>(), st)))
    return rc;
  if (Rt > 0 && ((rc = d2h(h_mean_out, om.p, sizeof(double) * Rt, st)) ||
                 (rc = d2h(h_var_out, ov.p, sizeof(double) * Rt, st))))
    return rc;
  if (h_info && (rc = d2h(h_info, di.p, sizeof(int) * B, st))) return rc;
  return sync(st);
}

// ------------------------------------------------------------------ train_new_gp (MG:84-104)
int magma_train_new_gp(magma_ctx* c, int n, const double* h_t, const double* h_y, const double* h_mean,
                       const double* h_cov, int64_t ld_cov, const double* ini3, const double* opt4,
                       double* h_out3, int* h_status) {
  if (!c || !h_t || !h_y || !h_mean || !ini3 || !h_out3) return fail(2, "train_new_gp: NULL argument");
  if (n < 1 || n > MAGMA_NSMALL_MAX) return fail(2, "train_new_gp: n must be in 1..64");
  if (int rc = set_device(c)) return rc;
  const Opt o(opt4);
  // one individual whose grid is its own timestamps: mean / new_cov given on that grid
  std::vector<int64_t> off = {0, n};
  magma_dataset* d = nullptr;
  if (int rc = magma_dataset_create(c, &d, 1, 1, 0, off.data(), h_t, h_y, n, h_t)) return rc;
  std::unique_ptr<magma_dataset, void (*)(magma_dataset*)> guard(d, magma_dataset_destroy);
  std::vector<double> zero;
  if (!h_cov) {
    zero.assign((size_t)n * n, 0.0);
    h_cov = zero.data();
    ld_cov = n;
  }
  if (int rc = magma_set_mean_cov(c, d, h_mean, h_cov, ld_cov)) return rc;
  hipStream_t st = c->main;
  if (int rc = ensure_cpack(d, st)) return rc;
  double x[3], f = 0.0;
  int status = 0;
  if (int rc = magma_indiv_mstep(1, 1, d->nmax, d->off.as<int64_t>(), d->t.as<double>(), d->y.as<double>(),
                                 d->idx.as<int>(), d->mean.as<double>(), d->cpack.as<double>(), ini3, o.m,
                                 o.factr, o.pgtol, o.maxit, d->theta.as<double>(), d->sel.as<int>(),
                                 d->f.as<double>(), d->g.as<double>(), d->info.as<int>(), x, &f, nullptr,
                                 nullptr, nullptr, &status, nullptr, nullptr, st))
    return rc;
  const bool bad = std::isnan(x[0]) || std::isnan(x[1]) || std::isnan(x[2]);  // MG:97-101
  for (int q = 0; q < 3; ++q) h_out3[q] = bad ? ini3[q] : x[q];
  if (h_status) *h_status = status;
  return 0;
}

}  // extern "C"
