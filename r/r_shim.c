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

/* ---- the R-named helpers on explicit arguments (kern_to_cov, kern_to_inv, dmvnorm, pred_gp) ---- */
static int pad_vec(double **dst, const double *src, int n, int Np) { /* device copy, zero padded */
  double *tmp = (double *)calloc(Np, sizeof(double));
  if (!tmp) return 1;
  memcpy(tmp, src, sizeof(double) * n);
  int rc = up((void **)dst, tmp, sizeof(double) * Np);
  if (!rc) rc = magma_stream_sync(NULL);
  free(tmp);
  return rc;
}

/* kern_to_cov / kern_to_inv, vector branch (CF:66-81, 121-138): t sorted ascending, theta3 =
 * (theta1, theta2, sigma). invert = FALSE: K (n x n); TRUE: list(K^-1, info) -- info != 0: not
 * positive definite, the caller takes the reference's solve() / ginv() body. Any n. */
SEXP magmahip_kern(SEXP t, SEXP theta3, SEXP invert) {
  const int n = LENGTH(t), Np = magma_pad(n), inv = asLogical(invert);
  double *g = NULL, *A = NULL, *out_d = NULL, *scal = NULL;
  void *work = NULL;
  int hinfo = 0;
  SEXP K = PROTECT(allocMatrix(REALSXP, n, n));
  int rc = pad_vec(&g, REAL(t), n, Np);
  if (!rc) rc = magma_dev_malloc((void **)&A, sizeof(double) * (size_t)Np * Np);
  if (!rc) rc = magma_dev_malloc((void **)&out_d, sizeof(double) * (size_t)n * n);
  if (!rc) rc = magma_kern_to_cov_grid(n, g, REAL(theta3)[0], REAL(theta3)[1], REAL(theta3)[2], A, NULL);
  if (!rc && inv) {
    rc = magma_dev_malloc((void **)&scal, 64);
    if (!rc) rc = magma_dev_malloc(&work, magma_workspace_bytes(n));
    if (!rc) rc = magma_spd_inverse(n, A, scal, (int *)(scal + 1), work, NULL);
    if (!rc) rc = magma_memcpy_d2h(&hinfo, scal + 1, sizeof(int), NULL);
  }
  if (!rc) rc = magma_copy_unpad(n, A, out_d, n, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(K), out_d, sizeof(double) * (size_t)n * n, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  if (g) magma_dev_free(g);
  if (A) magma_dev_free(A);
  if (out_d) magma_dev_free(out_d);
  if (scal) magma_dev_free(scal);
  if (work) magma_dev_free(work);
  if (rc) {
    UNPROTECT(1);
    Rf_error("magmahip: %s", magma_last_error());
  }
  if (!inv) {
    UNPROTECT(1);
    return K;
  }
  SEXP out = PROTECT(allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, K);
  SET_VECTOR_ELT(out, 1, ScalarInteger(hinfo));
  UNPROTECT(2);
  return out;
}

/* kern_to_cov / kern_to_inv, tibble branch (CF:85-112, 141-170) for individuals with n_i <= 64:
 * off (M + 1, 0-based), t (rows sorted within an individual), theta_i 3 x M or length 3.
 * Returns list(values: all K_i (or K_i^-1) column-major one after the other, info[M]). */
SEXP magmahip_kern_batched(SEXP off, SEXP t, SEXP theta_i, SEXP invert) {
  const int M = LENGTH(off) - 1, tot = LENGTH(t), inv = asLogical(invert);
  const int stride = LENGTH(theta_i) == 3 ? 0 : 3;
  int64_t *off64 = (int64_t *)malloc(sizeof(int64_t) * (M + 1));
  int64_t *off2 = (int64_t *)malloc(sizeof(int64_t) * (M + 1));
  if (!off64 || !off2) {
    free(off64);
    free(off2);
    Rf_error("magmahip: out of memory");
  }
  int nmax = 1;
  off2[0] = 0;
  for (int i = 0; i <= M; ++i) off64[i] = INTEGER(off)[i];
  for (int i = 0; i < M; ++i) {
    const int64_t n = off64[i + 1] - off64[i];
    if (n > nmax) nmax = (int)n;
    off2[i + 1] = off2[i] + n * n;
  }
  const int64_t n2 = off2[M];
  SEXP vals = PROTECT(allocVector(REALSXP, (R_xlen_t)n2));
  SEXP info = PROTECT(allocVector(INTSXP, M));
  memset(INTEGER(info), 0, sizeof(int) * M);
  void *d_off = NULL, *d_off2 = NULL, *d_t = NULL, *d_th = NULL, *d_out = NULL, *d_ld = NULL, *d_info = NULL;
  int rc = nmax > 64 ? 1 : 0;
  if (rc) Rf_error("magmahip: the batched builders take at most 64 observations per individual");
  rc = up(&d_off, off64, sizeof(int64_t) * (M + 1));
  if (!rc) rc = up(&d_off2, off2, sizeof(int64_t) * (M + 1));
  if (!rc) rc = up(&d_t, REAL(t), sizeof(double) * tot);
  if (!rc) rc = up(&d_th, REAL(theta_i), sizeof(double) * LENGTH(theta_i));
  if (!rc) rc = magma_dev_malloc(&d_out, sizeof(double) * (size_t)(n2 > 0 ? n2 : 1));
  if (!rc && inv) rc = magma_dev_malloc(&d_ld, sizeof(double) * (M > 0 ? M : 1));
  if (!rc && inv) rc = magma_dev_malloc(&d_info, sizeof(int) * (M > 0 ? M : 1));
  if (!rc && !inv)
    rc = magma_kern_to_cov_batched(M, d_off, d_off2, d_t, d_th, stride, d_out, NULL);
  if (!rc && inv)
    rc = magma_kern_to_inv_batched(M, nmax, d_off, d_off2, d_t, d_th, stride, d_out, d_ld, d_info, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(vals), d_out, sizeof(double) * (size_t)n2, NULL);
  if (!rc && inv) rc = magma_memcpy_d2h(INTEGER(info), d_info, sizeof(int) * M, NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  void *fr[] = {d_off, d_off2, d_t, d_th, d_out, d_ld, d_info};
  for (size_t i = 0; i < sizeof(fr) / sizeof(fr[0]); ++i)
    if (fr[i]) magma_dev_free(fr[i]);
  free(off64);
  free(off2);
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 2));
  SET_VECTOR_ELT(out, 0, vals);
  SET_VECTOR_ELT(out, 1, info);
  UNPROTECT(3);
  return out;
}

/* dmvnorm(x, mu, sigma = INVERSE covariance, log = TRUE) (CF:9-34): c(value, info). */
SEXP magmahip_dmvnorm(SEXP x, SEXP mu, SEXP inv_sigma) {
  const int n = LENGTH(x), Np = magma_pad(n);
  double *dx = NULL, *dmu = NULL, *dS = NULL, *tmp = NULL, hout = 0.0;
  void *work = NULL;
  int hinfo = 0;
  double *mu_full = (double *)malloc(sizeof(double) * n);
  if (!mu_full) Rf_error("magmahip: out of memory");
  for (int i = 0; i < n; ++i) mu_full[i] = REAL(mu)[LENGTH(mu) == 1 ? 0 : i];
  int rc = pad_vec(&dx, REAL(x), n, Np);
  if (!rc) rc = pad_vec(&dmu, mu_full, n, Np);
  free(mu_full);
  if (!rc) rc = up((void **)&tmp, REAL(inv_sigma), sizeof(double) * (size_t)n * n);
  if (!rc) rc = magma_dev_malloc((void **)&dS, sizeof(double) * (size_t)Np * Np);
  if (!rc) rc = magma_copy_pad(n, tmp, n, dS, 0, NULL);
  if (!rc) rc = magma_dev_malloc(&work, magma_workspace_bytes(n));
  if (!rc) rc = magma_dmvnorm(n, dx, dmu, dS, work, &hout, &hinfo, NULL);
  void *fr[] = {dx, dmu, dS, tmp, work};
  for (size_t i = 0; i < sizeof(fr) / sizeof(fr[0]); ++i)
    if (fr[i]) magma_dev_free(fr[i]);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  SEXP out = PROTECT(allocVector(REALSXP, 2));
  REAL(out)[0] = hout;
  REAL(out)[1] = (double)hinfo;
  UNPROTECT(1);
  return out;
}

/* pred_gp (MG:172-180) on explicit arguments: tn / yn / mean_obs (n <= 64, sorted by tn), ts /
 * mean_pred (T, sorted), cov_mu N x N with idx_n / idx_t = 0-based positions of tn / ts in its
 * dimnames (cov NULL: plain GP regression, MG:156-160), theta3 = (theta1, theta2, sigma).
 * list(Mean, Var, info). */
SEXP magmahip_pred_gp(SEXP tn, SEXP yn, SEXP mean_obs, SEXP idx_n, SEXP ts, SEXP mean_pred, SEXP idx_t,
                      SEXP cov, SEXP theta3) {
  const int n = LENGTH(tn), T = LENGTH(ts);
  const int have_cov = !isNull(cov);
  const int N = have_cov ? INTEGER(getAttrib(cov, R_DimSymbol))[0] : 0;
  void *d_tn = NULL, *d_yn = NULL, *d_mo = NULL, *d_in = NULL, *d_ts = NULL, *d_mp = NULL, *d_it = NULL;
  void *d_cov = NULL, *d_tmp = NULL, *d_th = NULL, *d_m = NULL, *d_v = NULL, *d_info = NULL;
  SEXP pm = PROTECT(allocVector(REALSXP, T));
  SEXP pv = PROTECT(allocVector(REALSXP, T));
  int pinfo = 0;
  int rc = up(&d_tn, REAL(tn), sizeof(double) * n);
  if (!rc) rc = up(&d_yn, REAL(yn), sizeof(double) * n);
  if (!rc) rc = up(&d_mo, REAL(mean_obs), sizeof(double) * n);
  if (!rc) rc = up(&d_ts, REAL(ts), sizeof(double) * T);
  if (!rc) rc = up(&d_mp, REAL(mean_pred), sizeof(double) * T);
  if (!rc) rc = up(&d_th, REAL(theta3), sizeof(double) * 3);
  if (!rc && have_cov) {
    rc = up(&d_in, INTEGER(idx_n), sizeof(int) * n);
    if (!rc) rc = up(&d_it, INTEGER(idx_t), sizeof(int) * T);
    if (!rc) rc = up(&d_tmp, REAL(cov), sizeof(double) * (size_t)N * N);
    if (!rc) rc = magma_dev_malloc(&d_cov, sizeof(double) * (size_t)magma_pad(N) * magma_pad(N));
    if (!rc) rc = magma_copy_pad(N, d_tmp, N, d_cov, 0, NULL);
  }
  if (!rc) rc = magma_dev_malloc(&d_m, sizeof(double) * (T > 0 ? T : 1));
  if (!rc) rc = magma_dev_malloc(&d_v, sizeof(double) * (T > 0 ? T : 1));
  if (!rc) rc = magma_dev_malloc(&d_info, sizeof(int));
  if (!rc) rc = magma_memset_zero(d_info, sizeof(int), NULL);
  if (!rc)
    rc = magma_pred_gp(n, d_tn, d_yn, d_mo, d_in, T, d_ts, d_mp, d_it, have_cov ? d_cov : NULL, N, d_th,
                       d_m, d_v, d_info, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(pm), d_m, sizeof(double) * T, NULL);
  if (!rc) rc = magma_memcpy_d2h(REAL(pv), d_v, sizeof(double) * T, NULL);
  if (!rc) rc = magma_memcpy_d2h(&pinfo, d_info, sizeof(int), NULL);
  if (!rc) rc = magma_stream_sync(NULL);
  void *fr[] = {d_tn, d_yn, d_mo, d_in, d_ts, d_mp, d_it, d_cov, d_tmp, d_th, d_m, d_v, d_info};
  for (size_t i = 0; i < sizeof(fr) / sizeof(fr[0]); ++i)
    if (fr[i]) magma_dev_free(fr[i]);
  if (rc) {
    UNPROTECT(2);
    Rf_error("magmahip: %s", magma_last_error());
  }
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, pm);
  SET_VECTOR_ELT(out, 1, pv);
  SET_VECTOR_ELT(out, 2, ScalarInteger(pinfo)); /* != 0: use the reference's solve() / ginv() body */
  UNPROTECT(3);
  return out;
}

/* e_step / posterior_mu body through the native driver (any n_i: individuals with more than 64
 * observations take the dense route inside the library): list(mean, cov, info). */
SEXP magmahip_e_step_native(SEXP hx, SEXP theta0, SEXP pen, SEXP theta_i, SEXP m0) {
  native *h = get_native(hx);
  int info = 0;
  int rc = magma_e_step(h->ctx, h->ds, REAL(theta0), REAL(theta_i), LENGTH(theta_i) == 3 ? 0 : 3,
                        REAL(pen)[0], REAL(m0), LENGTH(m0), &info, NULL);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  SEXP par = PROTECT(native_param(h));
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SET_VECTOR_ELT(out, 0, VECTOR_ELT(par, 0));
  SET_VECTOR_ELT(out, 1, VECTOR_ELT(par, 1));
  SET_VECTOR_ELT(out, 2, ScalarInteger(info));
  UNPROTECT(2);
  return out;
}

/* logL_monitoring (CF:236-264) through the native driver against a hyper-posterior given by the
 * caller (any n_i). */
SEXP magmahip_logl_monitoring_native(SEXP hx, SEXP mean, SEXP cov, SEXP m0, SEXP theta0, SEXP theta_i) {
  native *h = get_native(hx);
  double out = 0.0;
  int rc = magma_set_m0(h->ctx, h->ds, REAL(m0), LENGTH(m0));
  if (!rc) rc = magma_set_mean_cov(h->ctx, h->ds, REAL(mean), REAL(cov), h->N);
  if (!rc)
    rc = magma_logl_monitoring(h->ctx, h->ds, REAL(theta0), REAL(theta_i), LENGTH(theta_i) == 3 ? 0 : 3, &out);
  if (rc) Rf_error("magmahip: %s", magma_last_error());
  return ScalarReal(out);
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
    {"magmahip_kern", (DL_FUNC)&magmahip_kern, 3},
    {"magmahip_kern_batched", (DL_FUNC)&magmahip_kern_batched, 4},
    {"magmahip_dmvnorm", (DL_FUNC)&magmahip_dmvnorm, 3},
    {"magmahip_pred_gp", (DL_FUNC)&magmahip_pred_gp, 9},
    {"magmahip_e_step_native", (DL_FUNC)&magmahip_e_step_native, 5},
    {"magmahip_logl_monitoring_native", (DL_FUNC)&magmahip_logl_monitoring_native, 6},
    {NULL, NULL, 0}};

void R_init_magmahip(DllInfo *dll) {
  R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
