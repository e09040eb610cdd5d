"""CPU ORACLE — test infrastructure, NOT product code.

NumPy restatement, statement by statement, of the reference's multi-task-GP EM hot path
(`/root/reference/Computing_functions.R` = "CF", `/root/reference/MAGMA.R` = "MG").
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product package `magma_amd` never does.

Pinning: `posterior_mu` + `pred_gp` of this file reproduce the reference's
own stored results (`Simulations/Results/res_{mu,pred}_*.csv`) from its stored trained models
(`Simulations/Training/*.rds`) and datasets (`Simulations/Data/*.csv`) — see
`tests/test_oracle_kat.py` and the fixtures under `tests/golden/`. Functions no reference
fixture reaches (logL_*, gr_*, m_step, training trajectory, train_new_gp) are pinned by
finite differences and the E-step == posterior_mu identity only: for those, parity is
"unpinned" against the R implementation itself (R is not installed in this image).

Third-party arithmetic restated here (not under /root/reference, no version lock file; the
.rds header says R 3.6.3):
  * base R `solve()` -> LAPACK dgesv: `numpy.linalg.inv` (dgesv against identity).
  * base R `determinant()` -> LAPACK dgetrf: `numpy.linalg.slogdet`.
  * `MASS::ginv` fallback -> `numpy.linalg.pinv` (SVD, tol sqrt(eps)); parity unpinned.
  * `optimr::opm(method="L-BFGS-B")` -> `stats::optim` L-BFGS-B (factr 1e7, lmm 5, maxit 100,
    pgtol 0): restated with scipy's L-BFGS-B at the same settings; parity unpinned (only the
    optimiser's final outputs are stored in the .rds files).

Data conventions: a "db" is a dict of equal-length NumPy arrays `ID`, `Timestamp`, `Output`
(the tibble columns the reference requires). Hyper-parameters: `hp = {'theta_0': array(2),
'theta_i': {id: array(3)}}` keyed in `unique(db$ID)` order. Matrices carry their row/col
timestamp labels next to them (R uses dimnames `paste0('X', t)`; here exact float64 match).
"""
import math
import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------- helpers
def unique_in_order(a):
    """R's unique(): first-occurrence order."""
    _, first = np.unique(a, return_index=True)
    return np.asarray(a)[np.sort(first)]


def _solve_or_ginv(mat):
    """`tryCatch(solve(mat), error = function(e){MASS::ginv(mat)})` (CF:133,161,177,285,387;
    MG:123,173). solve() errors on an exactly singular LU or rcond < eps."""
    try:
        inv = np.linalg.inv(mat)
        if not np.all(np.isfinite(inv)):
            raise np.linalg.LinAlgError("non-finite inverse")
        # R: "system is computationally singular: reciprocal condition number = ..." when
        # rcond < .Machine$double.eps (La_solve: dgecon's 1-norm estimate; here the 1-norm
        # condition number of the computed inverse itself)
        if 1.0 / (np.abs(mat).sum(axis=0).max() * np.abs(inv).sum(axis=0).max()) < np.finfo(float).eps:
            raise np.linalg.LinAlgError("computationally singular")
        return inv
    except np.linalg.LinAlgError:
        return np.linalg.pinv(mat, rcond=np.sqrt(np.finfo(float).eps))


def mat_dist(x, y):
    """CF:36-39 — squared distances between all pairs of scalars in x and y."""
    x = np.asarray(x, float)
    y = np.asarray(y, float)
    return (x[:, None] - y[None, :]) ** 2


# --------------------------------------------------------------------------- kernels
def kernel(mat, theta):
    """CF:42-49 — SE kernel on squared distances: exp(theta1 - exp(-theta2)/2 * mat)."""
    return np.exp(theta[0] - np.exp(-theta[1]) / 2.0 * mat)


kernel_mu = kernel  # CF:51-57, identical body


def deriv_hp1(mat, theta):
    """CF:267-270."""
    return np.exp(theta[0] - np.exp(-theta[1]) / 2.0 * mat)


def deriv_hp2(mat, theta):
    """CF:272-277."""
    grad = 0.5 * np.exp(-theta[1]) * mat
    return np.exp(theta[0]) * grad * np.exp(-grad)


def _cov_vector(x, kern, theta, sigma):
    """Vector branch of kern_to_cov (CF:66-81): sort, off-diagonal = kern(dist^2)
    (`as.matrix.dist` has a zero diagonal), diagonal = sigma^2 + exp(theta1)."""
    x = np.sort(np.asarray(x, float).ravel())
    n = x.size
    if n == 1:
        return x, np.array([[sigma ** 2 + math.exp(theta[0])]])
    mat = kern(mat_dist(x, x), theta)
    np.fill_diagonal(mat, 0.0)
    mat = mat + np.diag(np.full(n, sigma ** 2 + math.exp(theta[0])))
    return x, mat


def kern_to_cov(x, kern=kernel, theta=(1.0, 0.5), sigma=0.2):
    """CF:59-113. Vector -> (labels, matrix). db dict -> {id: (labels, matrix)} with
    per-ID theta[[i]][1:3] or a common numeric theta + sigma (CF:89-93)."""
    if not isinstance(x, dict):
        return _cov_vector(x, kern, theta, sigma)
    out = {}
    for i in unique_in_order(x["ID"]):
        th = theta[i] if isinstance(theta, dict) else np.append(np.asarray(theta, float)[:2], sigma)
        indiv = np.sort(x["Timestamp"][x["ID"] == i])
        out[i] = _cov_vector(indiv, kern, th[:2], th[2])
    return out


def kern_to_inv(x, kern=kernel, theta=(1.0, 0.5), sigma=0.2):
    """CF:115-171 — same build, then solve()/ginv."""
    if not isinstance(x, dict):
        lab, mat = _cov_vector(x, kern, theta, sigma)
        return lab, _solve_or_ginv(mat)
    out = {}
    for i in unique_in_order(x["ID"]):
        th = theta[i] if isinstance(theta, dict) else np.append(np.asarray(theta, float)[:2], sigma)
        indiv = np.sort(x["Timestamp"][x["ID"] == i])
        lab, mat = _cov_vector(indiv, kern, th[:2], th[2])
        out[i] = (lab, _solve_or_ginv(mat))
    return out


# --------------------------------------------------------------------------- densities
def dmvnorm(x, mu, inv_Sigma, log=False):
    """CF:9-34 — takes the INVERSE covariance; log-det through LU of the inverse."""
    x = np.asarray(x, float).ravel()
    mu = np.asarray(mu, float).ravel()
    n = mu.size
    z = x - mu
    logdetS = -np.linalg.slogdet(inv_Sigma)[1]
    ssq = z @ inv_Sigma @ z
    loglik = -(n * LOG_2PI + logdetS + ssq) / 2.0
    return loglik if log else math.exp(loglik)


def _block(labels, mat, t):
    """`mat[paste0('X', t), paste0('X', t)]` — name-indexed block (exact float match)."""
    idx = np.searchsorted(labels, t)
    if np.any(idx >= labels.size) or np.any(labels[np.minimum(idx, labels.size - 1)] != t):
        raise KeyError("subscript out of bounds")  # R errors on a missing dimname
    return mat[np.ix_(idx, idx)]


def _mean_at(mean_db, t):
    """`mean %>% filter(Timestamp %in% t) %>% pull(Output)` — rows of the mean tibble whose
    Timestamp is in t, in the tibble's own order."""
    keep = np.isin(mean_db["Timestamp"], t)
    return mean_db["Output"][keep]


# --------------------------------------------------------------------------- log-likelihoods
def logL_GP(hp, db, mean, kern, new_cov):
    """CF:174-179."""
    _, cov = kern_to_cov(db["Timestamp"], kern, hp[:2], hp[2])
    cov = cov + new_cov
    inv = _solve_or_ginv(cov)
    return -dmvnorm(db["Output"], mean, inv, log=True)


def logL_GP_mod(hp, db, mean, kern, new_cov, pen_diag=None):
    """CF:181-201."""
    n = db["Timestamp"].size
    if np.ndim(mean) == 0 or np.size(mean) == 1:
        mean = np.full(n, float(np.ravel(mean)[0]))
    sigma = hp[2] if len(hp) == 3 else pen_diag
    _, inv = kern_to_inv(db["Timestamp"], kern, hp[:2], sigma)
    LL_norm = -dmvnorm(db["Output"], mean, inv, log=True)
    cor_term = 0.5 * np.sum(inv * new_cov)
    return LL_norm + cor_term


def logL_GP_mod_common_hp(hp, db, mean, kern, new_cov):
    """CF:203-234. `new_cov` is (labels, matrix); `mean` a db dict on the union grid."""
    labels, cov = new_cov
    LL_norm = 0.0
    cor_term = 0.0
    t_i_old = None
    inv = None
    for i in unique_in_order(db["ID"]):
        sel = db["ID"] == i
        t_i = db["Timestamp"][sel]
        y_i = db["Output"][sel]
        if t_i_old is None or not np.array_equal(t_i, t_i_old):
            _, inv = kern_to_inv(t_i, kern, hp[:2], hp[2])
        LL_norm = LL_norm - dmvnorm(y_i, _mean_at(mean, t_i), inv, log=True)
        cor_term = cor_term + 0.5 * np.sum(inv * _block(labels, cov, t_i))
        t_i_old = t_i
    return LL_norm + cor_term


def logL_monitoring(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0):
    """CF:236-264. `cov_mu` is (labels, matrix)."""
    labels, cov = cov_mu
    pen_diag = np.mean([hp["theta_i"][i][2] for i in hp["theta_i"]])
    ll_0 = logL_GP_mod(hp["theta_0"], mean_mu, m_0, kern_0, cov, pen_diag=pen_diag)
    sum_ll_i = 0.0
    for i in unique_in_order(db["ID"]):
        sel = db["ID"] == i
        t_i = db["Timestamp"][sel]
        db_i = {"Timestamp": t_i, "Output": db["Output"][sel]}
        sum_ll_i += logL_GP_mod(hp["theta_i"][i], db_i, _mean_at(mean_mu, t_i), kern_i,
                                _block(labels, cov, t_i))
    return -ll_0 - sum_ll_i


# --------------------------------------------------------------------------- gradients
def _noiseless_and_d2(t, hp):
    """kern_to_cov(t, deriv_hp1, theta, sigma = 0) (CF:289,316,355) and
    as.matrix(deriv_hp2(dist(t)^2, theta)) (CF:296,323,362; zero diagonal)."""
    _, k1 = _cov_vector(t, deriv_hp1, hp[:2], 0.0)
    ts = np.sort(np.asarray(t, float))
    k2 = deriv_hp2(mat_dist(ts, ts), hp[:2])
    np.fill_diagonal(k2, 0.0)
    return k1, k2


def gr_GP(hp, db, mean, kern, new_cov):
    """CF:279-302."""
    y = db["Output"]
    t = db["Timestamp"]
    _, cov = kern_to_cov(t, kern, hp[:2], hp[2])
    cov = cov + new_cov
    inv = _solve_or_ginv(cov)
    prod_inv = inv @ (y - mean)
    cste_term = np.outer(prod_inv, prod_inv) - inv
    k1, k2 = _noiseless_and_d2(t, hp)
    g_1 = 0.5 * np.trace(cste_term @ k1)
    g_2 = 0.0 if t.size == 1 else 0.5 * np.trace(cste_term @ k2)
    g_3 = hp[2] * np.trace(cste_term)
    return -np.array([g_1, g_2, g_3])


def gr_GP_mod(hp, db, mean, kern, new_cov, pen_diag=None):
    """CF:304-332."""
    y = db["Output"]
    t = db["Timestamp"]
    n = t.size
    sigma = hp[2] if len(hp) == 3 else pen_diag
    _, inv = kern_to_inv(t, kern, hp[:2], sigma)
    prod_inv = inv @ (y - mean)
    cste_term = np.outer(prod_inv, prod_inv) + inv @ (new_cov @ inv - np.eye(n))
    k1, k2 = _noiseless_and_d2(t, hp)
    g_1 = 0.5 * np.trace(cste_term @ k1)
    g_2 = 0.0 if n == 1 else 0.5 * np.trace(cste_term @ k2)
    if len(hp) == 3:
        g_3 = hp[2] * np.trace(cste_term)
        return -np.array([g_1, g_2, g_3])
    return -np.array([g_1, g_2])


def gr_GP_mod_common_hp(hp, db, mean, kern, new_cov):
    """CF:334-368."""
    labels, cov = new_cov
    g = np.zeros(3)
    t_i_old = None
    inv = None
    for i in unique_in_order(db["ID"]):
        sel = db["ID"] == i
        t_i = db["Timestamp"][sel]
        y_i = db["Output"][sel]
        n = t_i.size
        if t_i_old is None or not np.array_equal(t_i, t_i_old):
            _, inv = kern_to_inv(t_i, kern, hp[:2], hp[2])
        prod_inv = inv @ (y_i - _mean_at(mean, t_i))
        cste_term = np.outer(prod_inv, prod_inv) + inv @ (_block(labels, cov, t_i) @ inv - np.eye(n))
        k1, k2 = _noiseless_and_d2(t_i, hp)
        g[0] += 0.5 * np.trace(cste_term @ k1)
        if n > 1:
            g[1] += 0.5 * np.trace(cste_term @ k2)
        g[2] += hp[2] * np.trace(cste_term)
        t_i_old = t_i
    return -g


# --------------------------------------------------------------------------- E-step
def update_inv(prior_inv, list_inv_i):
    """CF:440-455. prior_inv = (labels, matrix); only labels present in the prior are used."""
    labels, new_inv = prior_inv[0], prior_inv[1].copy()
    for lab_i, inv_i in list_inv_i.values():
        common = np.isin(lab_i, labels)
        pos = np.searchsorted(labels, lab_i[common])
        new_inv[np.ix_(pos, pos)] += inv_i[np.ix_(common, common)]
    return labels, new_inv


def update_mean(prior_mean, prior_inv, list_inv_i, list_value_i):
    """CF:457-478."""
    labels, inv0 = prior_inv
    pm = np.full(labels.size, float(np.ravel(prior_mean)[0])) if np.size(prior_mean) == 1 \
        else np.asarray(prior_mean, float)
    weighted_mean = inv0 @ pm
    for i, (lab_i, inv_i) in list_inv_i.items():
        weighted_i = inv_i @ list_value_i[i]
        common = np.isin(lab_i, labels)
        pos = np.searchsorted(labels, lab_i[common])
        weighted_mean[pos] += weighted_i[common]
    return weighted_mean


def _hyperposterior(db, t_grid, m_0, kern_0, kern_i, hp, pen_diag):
    """Shared body of e_step (CF:379-394) and posterior_mu (MG:114-132)."""
    inv_0 = kern_to_inv(t_grid, kern_0, hp["theta_0"], pen_diag)
    inv_i = kern_to_inv(db, kern_i, hp["theta_i"], 0)
    value_i = {i: db["Output"][db["ID"] == i] for i in unique_in_order(db["ID"])}
    labels, new_inv = update_inv(inv_0, inv_i)
    new_cov = _solve_or_ginv(new_inv)
    weighted_mean = update_mean(m_0, inv_0, inv_i, value_i)
    new_mean = new_cov @ weighted_mean
    return {"mean": {"Timestamp": labels, "Output": new_mean},
            "cov": (labels, new_cov),
            "pred_GP": {"Timestamp": labels, "Mean": new_mean, "Var": np.diag(new_cov).copy()}}


def e_step(db, m_0, kern_0, kern_i, hp):
    """CF:371-395. pen_diag = signed mean of sigma_i (CF:381)."""
    all_t = np.unique(db["Timestamp"])
    pen_diag = np.mean([hp["theta_i"][i][2] for i in hp["theta_i"]])
    return _hyperposterior(db, all_t, m_0, kern_0, kern_i, hp, pen_diag)


# --------------------------------------------------------------------------- M-step
def opm_lbfgsb(par, fn, gr):
    """Stand-in for `opm(par, fn, gr, method = "L-BFGS-B", control = list(kkt = FALSE))[1, ]`
    (CF:412-413): stats::optim defaults factr=1e7, lmm=5, maxit=100, pgtol=0, no bounds."""
    from scipy.optimize import minimize
    res = minimize(fn, np.asarray(par, float), jac=gr, method="L-BFGS-B",
                   options={"maxcor": 5, "ftol": 1e7 * np.finfo(float).eps, "gtol": 0.0,
                            "maxiter": 100, "maxls": 20})
    return res.x


def m_step(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp, optimizer=None,
           batch_optimizer=None):
    """CF:397-437. `optimizer(par, fn, gr) -> par` replaces opm for the theta_0 and
    common-HP problems (default: scipy L-BFGS-B). `batch_optimizer(par0[M,3], fn_list,
    gr_list) -> par[M,3]`, if given, replaces the M independent opm calls of CF:423-431
    (lets a test drive this oracle and the GPU path with one and the same optimiser code)."""
    opt = optimizer or opm_lbfgsb
    labels, covm = cov
    list_ID = unique_in_order(db["ID"])
    pen_diag = np.mean([2.0 * old_hp["theta_i"][i][2] for i in old_hp["theta_i"]])
    new_theta_0 = opt(old_hp["theta_0"],
                      lambda p: logL_GP_mod(p, mean, m_0, kern_0, covm, pen_diag),
                      lambda p: gr_GP_mod(p, mean, m_0, kern_0, covm, pen_diag))
    if common_hp:
        param = opt(old_hp["theta_i"][list_ID[0]],
                    lambda p: logL_GP_mod_common_hp(p, db, mean, kern_i, cov),
                    lambda p: gr_GP_mod_common_hp(p, db, mean, kern_i, cov))
        new_theta_i = {i: np.array(param, float) for i in list_ID}
    else:
        fns, grs, par0 = [], [], []
        for i in list_ID:
            sel = db["ID"] == i
            t_i = db["Timestamp"][sel]
            db_i = {"Timestamp": t_i, "Output": db["Output"][sel]}
            mean_i = _mean_at(mean, t_i)
            cov_i = _block(labels, covm, t_i)
            fns.append(lambda p, d=db_i, m=mean_i, c=cov_i: logL_GP_mod(p, d, m, kern_i, c))
            grs.append(lambda p, d=db_i, m=mean_i, c=cov_i: gr_GP_mod(p, d, m, kern_i, c))
            par0.append(np.asarray(old_hp["theta_i"][i], float))
        if batch_optimizer is not None:
            res = batch_optimizer(np.array(par0), fns, grs)
        else:
            res = [opt(p, f, g) for p, f, g in zip(par0, fns, grs)]
        new_theta_i = {i: np.array(r, float) for i, r in zip(list_ID, res)}
    return {"theta_0": np.array(new_theta_0, float), "theta_i": new_theta_i}


# --------------------------------------------------------------------------- EM driver
def training(db, prior_mean, ini_hp, kern_0, kern_i, common_hp=True, optimizer=None,
             batch_optimizer=None, n_loop_max=25, trace=None):
    """MG:15-82. Returns hp, convergence, param (last E-step). `trace`, if a list, receives
    one dict per EM iteration (logLL of old hp, of new hp, eps) — the per-iteration
    log-likelihood the reference prints at MG:56,65."""
    list_ID = unique_in_order(db["ID"])
    hp = {"theta_0": np.asarray(ini_hp["theta_0"], float),
          "theta_i": {i: np.asarray(ini_hp["theta_i"], float) for i in list_ID}}
    cv = False
    logLL_monitoring = -np.inf
    param = None
    for it in range(1, n_loop_max + 1):
        param = e_step(db, prior_mean, kern_0, kern_i, hp)
        new_hp = m_step(db, hp, param["mean"], param["cov"], kern_0, kern_i, prior_mean,
                        common_hp, optimizer, batch_optimizer)
        vals = [new_hp["theta_0"]] + list(new_hp["theta_i"].values())
        if any(np.any(np.isnan(v)) for v in vals):  # MG:46-51
            break
        new_logLL = logL_monitoring(hp, db, kern_i, kern_0, param["mean"], param["cov"], prior_mean)
        _ = new_logLL - logLL_monitoring  # MG:57-58 (warning only)
        logL_new_hp = logL_monitoring(new_hp, db, kern_i, kern_0, param["mean"], param["cov"],
                                      prior_mean)
        eps = (logL_new_hp - new_logLL) / abs(logL_new_hp)
        if trace is not None:
            trace.append({"iter": it, "logLL": new_logLL, "logL_new_hp": logL_new_hp, "eps": eps})
        if eps < 1e-3:  # MG:66-71
            if eps > 0:
                hp = new_hp
            cv = True
            break
        hp = new_hp
        logLL_monitoring = new_logLL
    return {"hp": hp, "convergence": cv, "param": param}


def train_new_gp(db, mean_mu, cov_mu, ini_hp_i, kern_i, optimizer=None):
    """MG:84-104."""
    opt = optimizer or opm_lbfgsb
    n = db["Timestamp"].size
    if isinstance(mean_mu, dict):
        mean = _mean_at(mean_mu, db["Timestamp"])
    else:
        mean = np.asarray(mean_mu, float).ravel()
    if mean.size == 1:
        mean = np.full(n, mean[0])
    if isinstance(cov_mu, tuple):
        new_cov = _block(cov_mu[0], cov_mu[1], db["Timestamp"])
    else:
        new_cov = 0.0
    new_hp = opt(np.asarray(ini_hp_i, float),
                 lambda p: logL_GP(p, db, mean, kern_i, new_cov),
                 lambda p: gr_GP(p, db, mean, kern_i, new_cov))
    if np.any(np.isnan(new_hp)):
        new_hp = np.asarray(ini_hp_i, float)
    return {"theta": new_hp[:2], "sigma": new_hp[2]}


# --------------------------------------------------------------------------- prediction
def posterior_mu(db, new_db, timestamps, m_0, kern_0, kern_i, hp, pen_diag=0.1):
    """MG:107-133. `pen_diag` defaults to the committed 0.1 (MG:116); the stored results of
    three of the four result files were produced with the commented-out variant
    `mean(sigma_i)`, so it is a parameter here; pass 'mean_sigma' for that."""
    t_pred = np.unique(np.concatenate([np.asarray(timestamps, float).ravel(),
                                       db["Timestamp"], new_db["Timestamp"]]))
    if isinstance(pen_diag, str):
        pen_diag = np.mean([hp["theta_i"][i][2] for i in hp["theta_i"]])
    return _hyperposterior(db, t_pred, m_0, kern_0, kern_i, hp, pen_diag)


def pred_gp(db, timestamps=None, mean_mu=0.0, cov_mu=None, kern=kernel, theta=(1.0, 0.2),
            sigma=0.2):
    """MG:135-181. `cov_mu` = (labels, matrix) or None; `mean_mu` scalar or db dict."""
    tn = db["Timestamp"]
    yn = db["Output"]
    if timestamps is None:
        timestamps = np.linspace(tn.min(), tn.max(), 500)
    timestamps = np.asarray(timestamps, float).ravel()
    if cov_mu is None:
        all_times = unique_in_order(np.concatenate([tn, timestamps]))
        labels = np.sort(all_times)
        cov_mu = (labels, np.zeros((labels.size, labels.size)))
    labels, covm = cov_mu
    if not isinstance(mean_mu, dict):
        mean_mu_obs = np.full(tn.size, float(np.ravel(mean_mu)[0]))
        mean_mu_pred = np.full(timestamps.size, float(np.ravel(mean_mu)[0]))
    else:
        mean_mu_obs = _mean_at(mean_mu, tn)
        mean_mu_pred = _mean_at(mean_mu, timestamps)
    i_n = np.searchsorted(labels, tn)
    i_t = np.searchsorted(labels, timestamps)
    _, k_nn = kern_to_cov(tn, kern, theta, sigma)
    cov_tn_tn = k_nn + covm[np.ix_(i_n, i_n)]
    inv_mat = _solve_or_ginv(cov_tn_tn)
    cov_tn_t = kern(mat_dist(tn, timestamps), theta) + covm[np.ix_(i_n, i_t)]
    _, k_tt = kern_to_cov(timestamps, kern, theta, sigma)
    cov_t_t = k_tt + covm[np.ix_(i_t, i_t)]
    mean = mean_mu_pred + cov_tn_t.T @ inv_mat @ (yn - mean_mu_obs)
    var = np.diag(cov_t_t - cov_tn_t.T @ inv_mat @ cov_tn_t)
    return {"Timestamp": timestamps, "Mean": mean, "Var": var.copy()}


def full_algo(db, new_db, timestamps, kern_i, common_hp=True, prior_mean=0.0, kern_0=None,
              list_hp=None, mu=None, ini_hp=None, hp_new_i=None, pen_diag=0.1, optimizer=None):
    """MG:203-242 (plot argument dropped)."""
    if list_hp is None:
        list_hp = training(db, prior_mean, ini_hp, kern_0, kern_i, common_hp)["hp"]
    if mu is None:
        mu = posterior_mu(db, new_db, timestamps, prior_mean, kern_0, kern_i, list_hp, pen_diag)
    if hp_new_i is None and common_hp:
        first = next(iter(list_hp["theta_i"].values()))
        hp_new_i = {"theta": first[:2], "sigma": first[2]}
    elif hp_new_i is None:
        hp_new_i = train_new_gp(new_db, mu["mean"], mu["cov"], ini_hp["theta_i"], kern_i, optimizer)
    pred = pred_gp(new_db, timestamps, mu["mean"], mu["cov"], kern_i, hp_new_i["theta"],
                   hp_new_i["sigma"])
    return {"Prediction": pred, "Mean_process": mu,
            "Hyperparameters": {"other_i": list_hp, "new_i": hp_new_i}}


# --------------------------------------------------------------------------- evaluation quirks
def eval_methods_algo(pred_mean, pred_var, db_test):
    """`eval_methods` Algo row (simulation_study.R:188-204) incl. R's vector recycling when
    `pred` is longer than `db_test`."""
    pred_mean = np.asarray(pred_mean, float)
    sd = np.sqrt(np.asarray(pred_var, float))
    obs = np.resize(np.asarray(db_test, float), pred_mean.size)  # recycling
    mse = np.mean(np.abs(pred_mean - obs) ** 2)
    nb_between = np.sum((pred_mean - 1.96 * sd < obs) & (obs < pred_mean + 1.96 * sd))
    ratio = nb_between / np.size(db_test) * 100.0
    return mse, ratio
