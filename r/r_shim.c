/* R `.Call` shim over the C ABI of libmagma_hip.so (include/magma_hip.h).
 *
 * NOT compiled or tested in this repository's CI: the build image has no R (no Rinternals.h).
 * Build where R exists:   R CMD SHLIB r_shim.c -L../magma_amd -lmagma_hip -o magmahip.so
 * and load it from the sourced scripts with dyn.load("magmahip.so") (the reference is
 * script based: MAGMA.R:12 sources Computing_functions.R).
 *
 * Two layers: the per-kernel "session" calls below (what `opm` drives per evaluation), and the
 * native-driver calls at the end (magmahip_training, magmahip_m_step, ...: ONE .Call per reference
 * function, the EM loop, the optimisers and the theta_0 side stream inside the library).
 *
 * Conventions: inputs are read-only REAL()/INTEGER() views; outputs are freshly allocated,
 * column-major; a "session" external pointer owns the device copy of one db (ragged layout +
 * union grid + dense workspace) and is released by a finalizer. C ABI status codes become
 * Rf_error() AFTER all C resources of the call are released (no longjmp over live state).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/magma_hip.h"

typedef struct {
  int M, N, Np, nmax;
  int64_t *off;          /* device */
  double *t, *y, *grid;  /* device */
  int *idx;              /* device */
  double *theta, *f, *g, *P, *w, *mean, *m0, *cpack;
  int *info;
  void *work;
} session;

static void sess_free(session *s) {
  if (!s) return;
  void *p[] = {s->off, s->t, s->y, s->grid, s->idx, s->theta, s->f, s->g, s->P, s->w,
               s->mean, s->m0, s->info, s->work, s->cpack};
  for (size_t i = 0; i < sizeof(p) / sizeof(p[0]); ++i)
    if (p[i]) magma_dev_free(p[i]);
  free(s);
}

static void sess_finalizer(SEXP x) {
  sess_free((session *)R_ExternalPtrAddr(x));
  R_ClearExternalPtr(x);
}

static session *get_sess(SEXP x) {
  session *s = (session *)R_ExternalPtrAddr(x);
  if (!s) Rf_error("magmahip: session already released");
  return s;
}

static int up(void **dst, const void *src, size_t bytes) {
  int rc = magma_dev_malloc(dst, bytes ? bytes : 8);
  if (rc == 0 && bytes) rc = magma_memcpy_h2d(*dst, src, bytes, NULL);
  return rc;
}

/* off: integer M+1 (0-based row offsets), t/y: double, idx: integer (0-based), grid: double N */
SEXP magmahip_session(SEXP off, SEXP t, SEXP y, SEXP idx, SEXP grid) {
  session *s = (session *)calloc(1, sizeof(session));
  if (!s) Rf_error("magmahip: out of memory");
  s->M = LENGTH(off) - 1;
  s->N = LENGTH(grid);
  s->Np = magma_pad(s->N);
  const int tot = LENGTH(t);
  int64_t *off64 = (int64_t *)malloc(sizeof(int64_t) * (s->M + 1));
  double *gridp = (double *)calloc(s->Np, sizeof(double));
  int rc = (off64 && gridp) ? 0 : 1;
  if (rc == 0) {
    s->nmax = 1;
    for (int i = 0; i <= s->M; ++i) off64[i] = INTEGER(off)[i];
    for (int i = 0; i < s->M; ++i)
      if (off64[i + 1] - off64[i] > s->nmax) s->nmax = (int)(off64[i + 1] - off64[i]);
    memcpy(gridp, REAL(grid), sizeof(double) * s->N);
    const size_t n2 = (size_t)s->Np * s->Np, Mb = s->M > 0 ? s->M : 1;
    rc = up((void **)&s->off, off64, sizeof(int64_t) * (s->M + 1));
    if (!rc) rc = up((void **)&s->t, REAL(t), sizeof(double) * tot);
    if (!rc) rc = up((void **)&s->y, REAL(y), sizeof(double) * tot);
    if (!rc) rc = up((void **)&s->idx, INTEGER(idx), sizeof(int) * tot);
    if (!rc) rc = up((void **)&s->grid, gridp, sizeof(double) * s->Np);
    if (!rc) rc = magma_dev_malloc((void **)&s->theta, sizeof(double) * 3 * Mb);
    if (!rc) rc = magma_dev_malloc((void **)&s->f, sizeof(double) * Mb);
    if (!rc) rc = magma_dev_malloc((void **)&s->g, sizeof(double) * 3 * Mb);
    if (!rc) rc = magma_dev_malloc((void **)&s->info, sizeof(int) * Mb);
    if (!rc) rc = magma_dev_malloc((void **)&s->P, sizeof(double) * n2);
    if (!rc) rc = magma_dev_malloc((void **)&s->w, sizeof(double) * s->Np);
    if (!rc) rc = magma_dev_malloc((void **)&s->mean, sizeof(double) * s->Np);
    if (!rc) rc = magma_dev_malloc((void **)&s->m0, sizeof(double) * s->Np);
    if (!rc) rc = magma_dev_malloc((void **)&s->cpack, sizeof(double) * Mb * magma_cpack_stride(s->nmax));
    if (!rc) rc = magma_dev_malloc(&s->work, magma_workspace_bytes(s->N));
    if (!rc) rc = magma_stream_sync(NULL);
  }
  free(off64);
  free(gridp);
  if (rc) {
    sess_free(s);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP ptr = PROTECT(R_MakeExternalPtr(s, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ptr, sess_finalizer, TRUE);
  UNPROTECT(1);
  return ptr;
}

static int put_vec_padded(session *s, double *dst, SEXP v) {
  double *tmp = (double *)calloc(s->Np, sizeof(double));
  if (!tmp) return 1;
  const int n = LENGTH(v);
  for (int i = 0; i < s->N; ++i) tmp[i] = REAL(v)[n == 1 ? 0 : i];
  int rc = magma_memcpy_h2d(dst, tmp, sizeof(double) * s->Np, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  free(tmp);
  return rc;
}

/* e_step / posterior_mu body (CF:371-395, MG:107-133). theta_i: 3 x M matrix or length 3. */
SEXP magmahip_e_step(SEXP sx, SEXP theta0, SEXP pen, SEXP theta_i, SEXP m0) {
  session *s = get_sess(sx);
  const int stride = LENGTH(theta_i) == 3 ? 0 : 3;
  const size_t n2 = (size_t)s->Np * s->Np;
  int hinfo = 0;
  int rc = magma_memcpy_h2d(s->theta, REAL(theta_i), sizeof(double) * LENGTH(theta_i), NULL);
  if (!rc) rc = put_vec_padded(s, s->m0, m0);
  if (!rc) rc = magma_memset_zero(s->P, sizeof(double) * n2, NULL);
  if (!rc) rc = magma_memset_zero(s->w, sizeof(double) * s->Np, NULL);
  if (!rc) rc = magma_estep_scatter(s->M, s->nmax, s->off, s->t, s->y, s->idx, s->theta, stride,
                                    s->N, s->P, s->w, s->info, NULL);
  if (!rc) rc = magma_estep_finish(s->N, s->grid, REAL(theta0)[0], REAL(theta0)[1], REAL(pen)[0],
                                   s->m0, s->P, s->w, s->mean, s->work, 0, NULL, &hinfo, NULL);
  if (!rc && hinfo == 0)
    rc = magma_indiv_gather_cov(s->M, s->nmax, s->off, s->idx, s->P, s->N, s->cpack, NULL);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  SEXP mean = PROTECT(allocVector(REALSXP, s->N));
  SEXP cov = PROTECT(allocMatrix(REALSXP, s->N, s->N));
  double *tmp = NULL;
  rc = magma_dev_malloc((void **)&tmp, sizeof(double) * (size_t)s->N * s->N);
  if (!rc) rc = magma_copy_unpad(s->N, s->P, tmp, s->N, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(cov), tmp, sizeof(double) * (size_t)s->N * s->N, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(mean), s->mean, sizeof(double) * s->N, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  if (tmp) magma_dev_free(tmp);
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, mean);
  SET_VECTOR_ELT(out, 1, cov);
  SET_VECTOR_ELT(out, 2, ScalarInteger(hinfo)); /* != 0: fall back to solve()/ginv() in R */
  UNPROTECT(3);
  return out;
}

/* Load a hyper-posterior computed elsewhere (mean: N, cov: N x N on the session's grid). */
SEXP magmahip_set_posterior(SEXP sx, SEXP mean, SEXP cov) {
  session *s = get_sess(sx);
  double *tmp = NULL;
  int rc = put_vec_padded(s, s->mean, mean);
  if (!rc) rc = up((void **)&tmp, REAL(cov), sizeof(double) * (size_t)s->N * s->N);
  if (!rc) rc = magma_copy_pad(s->N, tmp, s->N, s->P, 0, NULL);
  if (!rc) rc = magma_indiv_gather_cov(s->M, s->nmax, s->off, s->idx, s->P, s->N, s->cpack, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  if (tmp) magma_dev_free(tmp);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  return R_NilValue;
}

/* logL_GP_mod / gr_GP_mod (mode 0) or logL_GP / gr_GP (mode 1) for every individual. */
SEXP magmahip_indiv_eval(SEXP sx, SEXP theta_i, SEXP want_grad, SEXP mode) {
  session *s = get_sess(sx);
  const int stride = LENGTH(theta_i) == 3 ? 0 : 3, wg = asLogical(want_grad);
  int rc = magma_memcpy_h2d(s->theta, REAL(theta_i), sizeof(double) * LENGTH(theta_i), NULL);
  if (!rc) rc = magma_indiv_eval(asInteger(mode), s->M, s->nmax, s->off, s->t, s->y, s->idx,
                                 s->theta, stride, s->mean, s->cpack, wg, NULL, 0, s->f, s->g,
                                 s->info, NULL);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  SEXP f = PROTECT(allocVector(REALSXP, s->M));
  SEXP g = PROTECT(allocMatrix(REALSXP, 3, s->M));
  SEXP info = PROTECT(allocVector(INTSXP, s->M));
  rc = magma_memcpy_d2h(REAL(f), s->f, sizeof(double) * s->M, NULL);
  if (!rc && wg) rc = magma_memcpy_d2h(REAL(g), s->g, sizeof(double) * 3 * s->M, NULL);
  if (!rc) rc = magma_memcpy_d2h(INTEGER(info), s->info, sizeof(int) * s->M, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  if (rc) {
    UNPROTECT(3);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, f);
  SET_VECTOR_ELT(out, 1, g);
  SET_VECTOR_ELT(out, 2, info);
  UNPROTECT(4);
  return out;
}

/* m_step's individual part (CF:419-432): one L-BFGS-B problem per individual, all advanced in
 * lock step by the library. theta_i: 3 x M start values. Returns list(par 3 x M, value M,
 * counts M, convergence-status M). */
SEXP magmahip_indiv_mstep(SEXP sx, SEXP theta_i, SEXP mode) {
  session *s = get_sess(sx);
  if (LENGTH(theta_i) != 3 * s->M) Rf_error("magmahip: theta_i must be 3 x M");
  int *sel = NULL;
  int rc = magma_dev_malloc((void **)&sel, sizeof(int) * (s->M > 0 ? s->M : 1));
  SEXP par = PROTECT(allocMatrix(REALSXP, 3, s->M));
  SEXP val = PROTECT(allocVector(REALSXP, s->M));
  SEXP nfev = PROTECT(allocVector(INTSXP, s->M));
  SEXP status = PROTECT(allocVector(INTSXP, s->M));
  if (!rc)
    rc = magma_indiv_mstep(asInteger(mode), s->M, s->nmax, s->off, s->t, s->y, s->idx, s->mean,
                           s->cpack, REAL(theta_i), 5, 1e7, 0.0, 100, s->theta, sel, s->f, s->g,
                           s->info, REAL(par), REAL(val), NULL, NULL, INTEGER(nfev),
                           INTEGER(status), NULL, NULL, NULL);
  if (sel) magma_dev_free(sel);
  if (rc) {
    UNPROTECT(4);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 4));
  SET_VECTOR_ELT(out, 0, par);
  SET_VECTOR_ELT(out, 1, val);
  SET_VECTOR_ELT(out, 2, nfev);
  SET_VECTOR_ELT(out, 3, status);
  UNPROTECT(5);
  return out;
}

/* logL_GP_mod + gr_GP_mod at theta_0: returns c(f, g1, g2, info). */
SEXP magmahip_theta0_eval(SEXP sx, SEXP theta0, SEXP pen, SEXP m0, SEXP want_grad) {
  session *s = get_sess(sx);
  double out3[3] = {0, 0, 0};
  int hinfo = 0;
  int rc = put_vec_padded(s, s->m0, m0);
  if (!rc) rc = magma_theta0_eval(s->N, s->grid, REAL(theta0)[0], REAL(theta0)[1], REAL(pen)[0],
                                  s->mean, s->m0, s->P, asLogical(want_grad), s->work, out3,
                                  &hinfo, NULL);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  SEXP out = PROTECT(allocVector(REALSXP, 4));
  REAL(out)[0] = out3[0];
  REAL(out)[1] = out3[1];
  REAL(out)[2] = out3[2];
  REAL(out)[3] = (double)hinfo;
  UNPROTECT(1);
  return out;
}

/* ---- the native driver: ONE .Call per reference function (magma_ctx / magma_dataset) ---- */
typedef struct {
  magma_ctx *ctx;
  magma_dataset *ds;
  int M, N;
} native;

static void native_free(native *h) {
  if (!h) return;
  if (h->ds) magma_dataset_destroy(h->ds);
  if (h->ctx) magma_ctx_destroy(h->ctx);
  free(h);
}

static void native_finalizer(SEXP x) {
  native_free((native *)R_ExternalPtrAddr(x));
  R_ClearExternalPtr(x);
}

static native *get_native(SEXP x) {
  native *h = (native *)R_ExternalPtrAddr(x);
  if (!h) Rf_error("magmahip: dataset already released");
  return h;
}

/* off: integer M+1 (0-based), t / y: double (rows of an individual contiguous, Timestamp
 * ascending), grid: double N (sorted unique). R is single-process: one context on device 0. */
SEXP magmahip_dataset(SEXP off, SEXP t, SEXP y, SEXP grid) {
  native *h = (native *)calloc(1, sizeof(native));
  if (!h) Rf_error("magmahip: out of memory");
  h->M = LENGTH(off) - 1;
  h->N = LENGTH(grid);
  int64_t *off64 = (int64_t *)malloc(sizeof(int64_t) * (h->M + 1));
  int rc = off64 ? 0 : 1;
  if (!rc) {
    for (int i = 0; i <= h->M; ++i) off64[i] = INTEGER(off)[i];
    rc = magma_ctx_create(&h->ctx, 0);
    if (!rc)
      rc = magma_dataset_create(h->ctx, &h->ds, h->M, h->M, 0, off64, REAL(t), REAL(y), h->N, REAL(grid));
  }
  free(off64);
  if (rc) {
    native_free(h);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP ptr = PROTECT(R_MakeExternalPtr(h, R_NilValue, R_NilValue));
  R_RegisterCFinalizerEx(ptr, native_finalizer, TRUE);
  UNPROTECT(1);
  return ptr;
}

static SEXP native_param(native *h) { /* list(mean N, cov N x N) of the resident hyper-posterior */
  SEXP mean = PROTECT(allocVector(REALSXP, h->N));
  SEXP cov = PROTECT(allocMatrix(REALSXP, h->N, h->N));
  int rc = magma_get_mean(h->ctx, h->ds, REAL(mean));
  if (!rc) rc = magma_get_cov(h->ctx, h->ds, REAL(cov), h->N);
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, mean);
  SET_VECTOR_ELT(out, 1, cov);
  UNPROTECT(3);
  return out;
}

/* training (MG:15-82) as one call: list(theta_0 (2), theta_i (3 x M), convergence, iterations,
 * trace (3 x n_loop_max: logLL(hp), logLL(new hp), eps per iteration), mean, cov). */
SEXP magmahip_training(SEXP hx, SEXP m0, SEXP ini_theta0, SEXP ini_theta_i, SEXP common_hp) {
  native *h = get_native(hx);
  const int nloop = 25; /* MG:23 */
  SEXP th0 = PROTECT(allocVector(REALSXP, 2));
  SEXP thi = PROTECT(allocMatrix(REALSXP, 3, h->M));
  SEXP trace = PROTECT(allocMatrix(REALSXP, 3, nloop));
  for (int k = 0; k < 3 * nloop; ++k) REAL(trace)[k] = NA_REAL;
  int cv = 0, its = 0;
  int rc = magma_training(h->ctx, h->ds, REAL(m0), LENGTH(m0), REAL(ini_theta0), REAL(ini_theta_i),
                          asLogical(common_hp), nloop, NULL, REAL(th0), REAL(thi), &cv, &its, REAL(trace));
  if (rc) {
    UNPROTECT(3);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP par = PROTECT(native_param(h));
  SEXP out = PROTECT(allocVector(VECSXP, 7));
  SET_VECTOR_ELT(out, 0, th0);
  SET_VECTOR_ELT(out, 1, thi);
  SET_VECTOR_ELT(out, 2, ScalarLogical(cv));
  SET_VECTOR_ELT(out, 3, ScalarInteger(its));
  SET_VECTOR_ELT(out, 4, trace);
  SET_VECTOR_ELT(out, 5, VECTOR_ELT(par, 0));
  SET_VECTOR_ELT(out, 6, VECTOR_ELT(par, 1));
  UNPROTECT(5);
  return out;
}

/* m_step (CF:397-437), both branches, against a hyper-posterior given by the caller:
 * list(theta_0 (2), theta_i (3 x M)). */
SEXP magmahip_m_step(SEXP hx, SEXP mean, SEXP cov, SEXP m0, SEXP theta0, SEXP theta_i, SEXP common_hp) {
  native *h = get_native(hx);
  if (LENGTH(theta_i) != 3 * h->M) Rf_error("magmahip: theta_i must be 3 x M");
  SEXP th0 = PROTECT(allocVector(REALSXP, 2));
  SEXP thi = PROTECT(allocMatrix(REALSXP, 3, h->M));
  int rc = magma_set_m0(h->ctx, h->ds, REAL(m0), LENGTH(m0));
  if (!rc) rc = magma_set_mean_cov(h->ctx, h->ds, REAL(mean), REAL(cov), h->N);
  if (!rc) rc = magma_m_step(h->ctx, h->ds, REAL(theta0), REAL(theta_i), asLogical(common_hp), NULL,
                             REAL(th0), REAL(thi));
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, th0);
  SET_VECTOR_ELT(out, 1, thi);
  UNPROTECT(3);
  return out;
}

/* train_new_gp (MG:84-104): t / y sorted by t, mean and cov block at t -> c(theta1, theta2, sigma) */
SEXP magmahip_train_new_gp(SEXP t, SEXP y, SEXP mean, SEXP cov, SEXP ini) {
  magma_ctx *ctx = NULL;
  const int n = LENGTH(t);
  SEXP out = PROTECT(allocVector(REALSXP, 3));
  int rc = magma_ctx_create(&ctx, 0);
  if (!rc)
    rc = magma_train_new_gp(ctx, n, REAL(t), REAL(y), REAL(mean), isNull(cov) ? NULL : REAL(cov), n,
                            REAL(ini), NULL, REAL(out), NULL);
  if (ctx) magma_ctx_destroy(ctx);
  if (rc) {
    UNPROTECT(1);
    Rf_error("magmahip: %s", magma_last_error());
  }
  UNPROTECT(1);
  return out;
}

/* posterior_mu (MG:107-133) + pred_gp (MG:135-181) for ONE new individual in one call, the
 * hyper-posterior never leaving the device: list(Mean, Var) at the (sorted) target timestamps. */
SEXP magmahip_posterior_pred(SEXP off, SEXP t, SEXP y, SEXP extra_t, SEXP theta0, SEXP theta_i,
                             SEXP pen, SEXP m0, SEXP tn, SEXP yn, SEXP ts, SEXP theta_new) {
  const int M = LENGTH(off) - 1, T = LENGTH(ts), n = LENGTH(tn);
  magma_ctx *ctx = NULL;
  magma_dataset *ds = NULL;
  int64_t *off64 = (int64_t *)malloc(sizeof(int64_t) * (M + 1));
  SEXP pm = PROTECT(allocVector(REALSXP, T));
  SEXP pv = PROTECT(allocVector(REALSXP, T));
  int rc = off64 ? 0 : 1, N = 0, info = 0, pinfo = 0;
  if (!rc) {
    for (int i = 0; i <= M; ++i) off64[i] = INTEGER(off)[i];
    rc = magma_ctx_create(&ctx, 0);
  }
  if (!rc)
    rc = magma_posterior_mu(ctx, &ds, M, M, 0, off64, REAL(t), REAL(y), LENGTH(extra_t), REAL(extra_t),
                            REAL(theta0), REAL(theta_i), LENGTH(theta_i) == 3 ? 0 : 3, REAL(pen)[0],
                            REAL(m0), &N, &info);
  if (!rc && info == 0) {
    const int64_t on[2] = {0, n}, ot[2] = {0, T};
    rc = magma_pred_gp_resident(ctx, ds, 1, on, REAL(tn), REAL(yn), ot, REAL(ts), REAL(theta_new), 0,
                                REAL(pm), REAL(pv), &pinfo);
  }
  free(off64);
  if (ds) magma_dataset_destroy(ds);
  if (ctx) magma_ctx_destroy(ctx);
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, pm);
  SET_VECTOR_ELT(out, 1, pv);
  SET_VECTOR_ELT(out, 2, ScalarInteger(info != 0 ? info : pinfo)); /* != 0: use the R bodies */
  UNPROTECT(3);
  return out;
}

static const R_CallMethodDef call_methods[] = {
    {"magmahip_dataset", (DL_FUNC)&magmahip_dataset, 4},
    {"magmahip_training", (DL_FUNC)&magmahip_training, 5},
    {"magmahip_m_step", (DL_FUNC)&magmahip_m_step, 7},
    {"magmahip_train_new_gp", (DL_FUNC)&magmahip_train_new_gp, 5},
    {"magmahip_posterior_pred", (DL_FUNC)&magmahip_posterior_pred, 12},
    {"magmahip_session", (DL_FUNC)&magmahip_session, 5},
    {"magmahip_e_step", (DL_FUNC)&magmahip_e_step, 5},
    {"magmahip_set_posterior", (DL_FUNC)&magmahip_set_posterior, 3},
    {"magmahip_indiv_eval", (DL_FUNC)&magmahip_indiv_eval, 4},
    {"magmahip_indiv_mstep", (DL_FUNC)&magmahip_indiv_mstep, 3},
    {"magmahip_theta0_eval", (DL_FUNC)&magmahip_theta0_eval, 5},
    {NULL, NULL, 0}};

void R_init_magmahip(DllInfo *dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
