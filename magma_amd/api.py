"""Host-side mirror of the reference's R function surface, running on the HIP library.

Same names, argument order and return shapes as MAGMA.R (MG) / Computing_functions.R (CF);
R is not available in this image, so the host language is Python (see INTEGRATION.md for the
`.Call` shim an R maintainer would add over the same C ABI). Conventions that replace R types:
  * a "db" tibble is a dict of equal-length arrays `ID`, `Timestamp`, `Output`;
  * `hp = {'theta_0': array(2), 'theta_i': {id: array(3)}}` (R: named list of vectors);
  * a matrix with `X<t>` dimnames is a pair `(labels, matrix)` with float64 labels;
  * `kern_0` / `kern_i` must be the SE kernel (`kernel` / `kernel_mu` below): it is the only
    kernel the reference has gradients for (CF:267-277).
"""
import ctypes as C
import numpy as np

from . import _lib
from .engine import Dataset, HipBackend, unique_in_order, _ptr
from .native import NativeEngine, engine_for_job


# ------------------------------------------------------------------ kernels (host, CF:36-57)
def mat_dist(x, y):
    x = np.asarray(x, float)
    y = np.asarray(y, float)
    return (x[:, None] - y[None, :]) ** 2


def kernel(mat, theta=(1.0, 0.5)):
    return np.exp(theta[0] - np.exp(-theta[1]) / 2.0 * np.asarray(mat, float))


def kernel_mu(mat, theta=(1.0, 0.5)):
    return np.exp(theta[0] - np.exp(-theta[1]) / 2.0 * np.asarray(mat, float))


def deriv_hp1(mat, theta=(1.0, 0.5)):
    """CF:267-270 (closure passed as `kern` to kern_to_cov by the gradient functions)."""
    return np.exp(theta[0] - np.exp(-theta[1]) / 2.0 * np.asarray(mat, float))


def deriv_hp2(mat, theta=(1.0, 0.5)):
    """CF:272-277."""
    grad = 0.5 * np.exp(-theta[1]) * np.asarray(mat, float)
    return np.exp(theta[0]) * grad * np.exp(-grad)


def _check_kern(*kerns):
    for k in kerns:
        if k is not None and k not in (kernel, kernel_mu):
            raise NotImplementedError("the HIP path implements the reference's SE kernel "
                                      "(`kernel` / `kernel_mu`) only")


def _hp_arrays(hp, ids):
    th0 = np.asarray(hp["theta_0"], dtype=np.float64).ravel()[:2]
    thi = np.array([np.asarray(hp["theta_i"][i], dtype=np.float64).ravel()[:3] for i in ids])
    return th0, thi


def _hp_dict(theta0, theta_i, ids):
    return {"theta_0": np.array(theta0), "theta_i": {i: np.array(theta_i[k]) for k, i in enumerate(ids)}}


def _as_db(x):
    return {k: np.asarray(v) for k, v in x.items()}


# ------------------------------------------------------------------ covariance builders
def _single_individual_backend(t, y=None, grid=None):
    t = np.sort(np.asarray(t, dtype=np.float64).ravel())
    ds = Dataset.from_arrays([0, t.size], t, np.zeros(t.size) if y is None else y, grid=grid)
    return ds, HipBackend(ds, 0, 1)


def kern_to_cov(x, kern=kernel, theta=(1.0, 0.5), sigma=0.2):
    """CF:59-113. Vector -> (labels, K); db -> {id: (labels, K_i)} (per-ID theta[[i]] or a
    common theta + sigma)."""
    _check_kern(kern)
    import torch
    lib = _lib.load()
    if not isinstance(x, dict):
        t = np.sort(np.asarray(x, dtype=np.float64).ravel())
        N = t.size
        Np = lib.magma_pad(N)
        g = torch.zeros(Np, dtype=torch.float64, device="cuda")
        g[:N] = torch.from_numpy(t)
        A = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call("magma_kern_to_cov_grid", N, _ptr(g), float(theta[0]), float(theta[1]),
                  float(sigma), _ptr(A), st)
        return t, A.view(Np, Np)[:N, :N].t().contiguous().cpu().numpy()
    return _batched_small(x, theta, sigma, inverse=False)


def kern_to_inv(x, kern=kernel, theta=(1.0, 0.5), sigma=0.2):
    """CF:115-171."""
    _check_kern(kern)
    import torch
    lib = _lib.load()
    if not isinstance(x, dict):
        t = np.sort(np.asarray(x, dtype=np.float64).ravel())
        N = t.size
        Np = lib.magma_pad(N)
        g = torch.zeros(Np, dtype=torch.float64, device="cuda")
        g[:N] = torch.from_numpy(t)
        A = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
        work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
        scal = torch.zeros(2, dtype=torch.float64, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call("magma_kern_to_cov_grid", N, _ptr(g), float(theta[0]), float(theta[1]),
                  float(sigma), _ptr(A), st)
        _lib.call("magma_spd_inverse", N, _ptr(A), _ptr(scal), _ptr(info), _ptr(work), st)
        if int(info.item()) != 0:
            raise np.linalg.LinAlgError(f"covariance not positive definite (info={int(info.item())})")
        return t, A.view(Np, Np)[:N, :N].t().contiguous().cpu().numpy()
    return _batched_small(x, theta, sigma, inverse=True)


def _batched_small(db, theta, sigma, inverse):
    import torch
    db = _as_db(db)
    ds = Dataset(db)
    if ds.nmax > 64:
        # individuals with more than 64 observations: the vector branch (dense kernels) each, the
        # others batched (CF:85-112, 141-170 make no such distinction)
        n = np.diff(ds.off)
        res, small_rows = {}, []
        for k, i in enumerate(ds.ids):
            sl = slice(ds.off[k], ds.off[k + 1])
            if n[k] > 64:
                th = np.asarray(theta[i], float) if isinstance(theta, dict) else np.append(
                    np.asarray(theta, float)[:2], sigma)
                fn = kern_to_inv if inverse else kern_to_cov
                res[i] = fn(ds.t[sl], kernel, th[:2], th[2])
            else:
                small_rows.append(np.arange(sl.start, sl.stop))
        if small_rows:
            rows = np.concatenate(small_rows)
            sub = {"ID": np.repeat(np.array(ds.ids, dtype=object), n)[rows], "Timestamp": ds.t[rows],
                   "Output": ds.y[rows]}
            res.update(_batched_small(sub, theta, sigma, inverse))
        return {i: res[i] for i in ds.ids}
    if isinstance(theta, dict):
        th = np.array([np.asarray(theta[i], float)[:3] for i in ds.ids])
        stride = 3
    else:
        th = np.append(np.asarray(theta, float)[:2], sigma)[None, :]
        stride = 0
    n = np.diff(ds.off)
    off2 = np.concatenate([[0], np.cumsum(n * n)]).astype(np.int64)
    dev = "cuda"
    d_off = torch.from_numpy(ds.off).to(dev)
    d_off2 = torch.from_numpy(off2).to(dev)
    d_t = torch.from_numpy(ds.t).to(dev)
    d_th = torch.from_numpy(th.reshape(-1).copy()).to(dev)
    out = torch.empty(int(off2[-1]), dtype=torch.float64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if inverse:
        ld = torch.empty(ds.M, dtype=torch.float64, device=dev)
        info = torch.empty(ds.M, dtype=torch.int32, device=dev)
        _lib.call("magma_kern_to_inv_batched", ds.M, ds.nmax, _ptr(d_off), _ptr(d_off2), _ptr(d_t),
                  _ptr(d_th), stride, _ptr(out), _ptr(ld), _ptr(info), st)
        # (info != 0: that matrix was answered by the library's solve() / ginv() slow path, the
        # reference's tryCatch at CF:161)
    else:
        _lib.call("magma_kern_to_cov_batched", ds.M, _ptr(d_off), _ptr(d_off2), _ptr(d_t),
                  _ptr(d_th), stride, _ptr(out), st)
    flat = out.cpu().numpy()
    res = {}
    for k, i in enumerate(ds.ids):
        ni = int(n[k])
        res[i] = (ds.t[ds.off[k]:ds.off[k + 1]].copy(),
                  flat[off2[k]:off2[k + 1]].reshape(ni, ni).T.copy())
    return res


# ------------------------------------------------------------------ density, E-step pieces
def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _padded_matrix(mat, N, Np):
    """Host [N, N] -> device Np x Np column-major with identity padding."""
    import torch
    c = np.eye(Np)
    c[:N, :N] = mat
    return torch.from_numpy(np.ascontiguousarray(c.T).reshape(-1)).cuda()


def dmvnorm(x, mu, inv_Sigma, log=False):
    """CF:9-34: Gaussian (log-)density at x given the INVERSE covariance (magma_dmvnorm)."""
    import torch
    lib = _lib.load()
    x = np.asarray(x, dtype=np.float64).ravel()
    n = x.size
    mu = np.asarray(mu, dtype=np.float64).ravel()
    mu = np.full(n, mu[0]) if mu.size == 1 else mu
    Np = lib.magma_pad(n)
    pad = lambda v: torch.from_numpy(np.concatenate([v, np.zeros(Np - n)])).cuda()  # noqa: E731
    d_x, d_mu = pad(x), pad(mu)
    d_inv = _padded_matrix(np.asarray(inv_Sigma, dtype=np.float64), n, Np)
    work = torch.empty((lib.magma_workspace_bytes(n) + 7) // 8, dtype=torch.float64, device="cuda")
    out, info = C.c_double(0.0), C.c_int(0)
    _lib.call("magma_dmvnorm", n, _ptr(d_x), _ptr(d_mu), _ptr(d_inv), _ptr(work), C.byref(out),
              C.byref(info), _stream())
    if info.value != 0:
        raise np.linalg.LinAlgError(f"inverse covariance not positive definite (info={info.value})")
    return out.value if log else float(np.exp(out.value))


def _explicit_blocks(labels, list_inv_i):
    """Ragged device layout of a named list of (labels_i, K_i^-1). A timestamp that is not in
    `labels` gets idx -1: the library drops it like `intersect(row.names(...))` (CF:451,474)."""
    import torch
    off, off2, idx, blocks = [0], [0], [], []
    for lab_i, inv_i in list_inv_i.values():
        lab_i = np.asarray(lab_i, float)
        pos = np.minimum(np.searchsorted(labels, lab_i), labels.size - 1)
        idx.append(np.where(labels[pos] == lab_i, pos, -1).astype(np.int32))
        blocks.append(np.ascontiguousarray(np.asarray(inv_i, float).T).reshape(-1))
        off.append(off[-1] + lab_i.size)
        off2.append(off2[-1] + lab_i.size ** 2)
    cat = lambda xs, dt: torch.from_numpy(  # noqa: E731
        np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)).cuda()
    return (len(blocks), torch.tensor(off, dtype=torch.int64).cuda(),
            torch.tensor(off2, dtype=torch.int64).cuda(), cat(idx, np.int32),
            cat(blocks, np.float64))


def update_inv(prior_inv, list_inv_i):
    """CF:440-455: prior_inv = (labels, K_0^-1), list_inv_i = {id: (labels_i, K_i^-1)} ->
    (labels, K_0^-1 + sum_i scatter(K_i^-1)) (magma_update_inv)."""
    lib = _lib.load()
    labels, inv0 = np.asarray(prior_inv[0], float), np.asarray(prior_inv[1], float)
    N = labels.size
    Np = lib.magma_pad(N)
    P = _padded_matrix(inv0, N, Np)
    M, off, off2, idx, Kinv = _explicit_blocks(labels, list_inv_i)
    _lib.call("magma_update_inv", M, _ptr(off), _ptr(off2), _ptr(idx), _ptr(Kinv), N, _ptr(P),
              _stream())
    return labels, P.view(Np, Np)[:N, :N].t().contiguous().cpu().numpy()


def update_mean(prior_mean, prior_inv, list_inv_i, list_value_i):
    """CF:457-478: K_0^-1 m_0 + sum_i scatter(K_i^-1 y_i) on the grid of prior_inv
    (magma_update_mean; the prior term is a host product of data the caller already holds)."""
    import torch
    lib = _lib.load()
    labels, inv0 = np.asarray(prior_inv[0], float), np.asarray(prior_inv[1], float)
    N = labels.size
    Np = lib.magma_pad(N)
    pm = np.asarray(prior_mean, float).ravel()
    pm = np.full(N, pm[0]) if pm.size == 1 else pm
    # K_0^-1 m_0 through the library's product kernel (m_0 as the first row of a 64-row operand)
    P = _padded_matrix(inv0, N, Np)
    B = torch.zeros(Np * 64, dtype=torch.float64, device="cuda")
    B.view(Np, 64)[:N, 0] = torch.from_numpy(pm).cuda()
    Cm = torch.empty(64 * Np, dtype=torch.float64, device="cuda")
    _lib.call("magma_gemm_nt", Np, 64, Np, 1.0, _ptr(P), Np, _ptr(B), 64, 0.0, _ptr(Cm), Np,
              _stream())
    w = Cm[:Np].clone()
    w[N:] = 0.0
    M, off, off2, idx, Kinv = _explicit_blocks(labels, list_inv_i)
    ys = [np.asarray(list_value_i[i], float).ravel() for i in list_inv_i]
    y = torch.from_numpy(np.concatenate(ys) if ys else np.zeros(0)).cuda()
    _lib.call("magma_update_mean", M, _ptr(off), _ptr(off2), _ptr(idx), _ptr(Kinv), _ptr(y), N,
              _ptr(w), _stream())
    return w[:N].cpu().numpy()


# ------------------------------------------------------------------ likelihoods / gradients
def _indiv_problem(db, mean, new_cov):
    """Single-individual engine for logL_GP_mod / gr_GP_mod / logL_GP / gr_GP with explicit
    mean vector and covariance block (as the reference passes them)."""
    t = np.asarray(db["Timestamp"], dtype=np.float64)
    y = np.asarray(db["Output"], dtype=np.float64)
    order = np.argsort(t, kind="stable")
    ds = Dataset.from_arrays([0, t.size], t[order], y[order])
    be = HipBackend(ds, 0, 1)
    mean = np.asarray(mean, dtype=np.float64).ravel()
    mean = np.full(t.size, mean[0]) if mean.size == 1 else mean[order]
    cov = np.zeros((t.size, t.size)) if np.ndim(new_cov) == 0 else np.asarray(new_cov, float)[np.ix_(order, order)]
    be.set_mean_cov(mean, cov)
    return be


def _theta0_problem(db, mean, new_cov):
    t = np.asarray(db["Timestamp"], dtype=np.float64)
    ds = Dataset.from_arrays([0, 0], np.zeros(0), np.zeros(0), grid=np.sort(t))
    be = HipBackend(ds, 0, 0)
    order = np.argsort(t, kind="stable")
    be.set_mean_cov(np.asarray(db["Output"], float)[order], np.asarray(new_cov, float)[np.ix_(order, order)])
    be.set_m0(mean if np.size(mean) == 1 else np.asarray(mean, float)[order])
    return be


def logL_GP_mod(hp, db, mean, kern, new_cov, pen_diag=None):
    """CF:181-201."""
    _check_kern(kern)
    hp = np.asarray(hp, dtype=np.float64).ravel()
    if hp.size == 3:
        f, _ = _indiv_problem(db, mean, new_cov).indiv_eval(hp[None, :], want_grad=False)
        return float(f[0])
    f, _, _ = _theta0_problem(db, mean, new_cov).theta0_eval(hp, pen_diag, want_grad=False)
    return float(f)


def gr_GP_mod(hp, db, mean, kern, new_cov, pen_diag=None):
    """CF:304-332."""
    _check_kern(kern)
    hp = np.asarray(hp, dtype=np.float64).ravel()
    if hp.size == 3:
        _, g = _indiv_problem(db, mean, new_cov).indiv_eval(hp[None, :], want_grad=True)
        return g[0]
    _, g, _ = _theta0_problem(db, mean, new_cov).theta0_eval(hp, pen_diag, want_grad=True)
    return g


def logL_GP(hp, db, mean, kern, new_cov):
    """CF:174-179."""
    _check_kern(kern)
    hp = np.asarray(hp, dtype=np.float64).ravel()
    f, _ = _indiv_problem(db, mean, new_cov).indiv_eval(hp[None, :], want_grad=False, mode=1)
    return float(f[0])


def gr_GP(hp, db, mean, kern, new_cov):
    """CF:279-302."""
    _check_kern(kern)
    hp = np.asarray(hp, dtype=np.float64).ravel()
    _, g = _indiv_problem(db, mean, new_cov).indiv_eval(hp[None, :], want_grad=True, mode=1)
    return g[0]


def _engine_with_posterior(db, mean, new_cov):
    """Engine over db whose grid is the labels of (labels, cov) with mean/cov loaded."""
    labels, cov = new_cov
    ds = Dataset(_as_db(db), extra_timestamps=labels)
    if ds.N != labels.size:
        raise KeyError("db has timestamps missing from the covariance's dimnames")
    eng = engine_for_job(ds)
    mvec = np.asarray(mean["Output"], float)[np.argsort(mean["Timestamp"], kind="stable")]
    eng.set_mean_cov(mvec, cov)
    return eng


def logL_GP_mod_common_hp(hp, db, mean, kern, new_cov):
    """CF:203-234."""
    _check_kern(kern)
    eng = _engine_with_posterior(db, mean, new_cov)
    f, _ = eng.sum_fi(np.asarray(hp, dtype=np.float64).ravel(), want_grad=False)
    return float(f)


def gr_GP_mod_common_hp(hp, db, mean, kern, new_cov):
    """CF:334-368."""
    _check_kern(kern)
    eng = _engine_with_posterior(db, mean, new_cov)
    _, g = eng.sum_fi(np.asarray(hp, dtype=np.float64).ravel(), want_grad=True)
    return g


def logL_monitoring(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0):
    """CF:236-264."""
    _check_kern(kern_i, kern_0)
    eng = _engine_with_posterior(db, mean_mu, cov_mu)
    eng.be.set_m0(m_0)
    th0, thi = _hp_arrays(hp, eng.ds.ids)
    return float(eng.logL_monitoring(th0, thi))


# ------------------------------------------------------------------ EM functions
def _param_from_engine(eng):
    grid = eng.ds.grid.copy()
    mean = eng.be.get_mean()
    cov = eng.be.get_cov()
    return {"mean": {"Timestamp": grid, "Output": mean},
            "cov": (grid, cov),
            "pred_GP": {"Timestamp": grid, "Mean": mean.copy(), "Var": np.diag(cov).copy()}}


def e_step(db, m_0, kern_0, kern_i, hp):
    """CF:371-395."""
    _check_kern(kern_0, kern_i)
    eng = engine_for_job(Dataset(_as_db(db)))
    th0, thi = _hp_arrays(hp, eng.ds.ids)
    info = eng.e_step(th0, thi, float(np.mean(thi[:, 2])), m_0)
    if info != 0:
        raise np.linalg.LinAlgError(f"hyper-posterior precision not positive definite ({info})")
    return _param_from_engine(eng)


def m_step(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp):
    """CF:397-437."""
    _check_kern(kern_0, kern_i)
    eng = _engine_with_posterior(db, mean, cov)
    eng.be.set_m0(m_0)
    th0, thi = _hp_arrays(old_hp, eng.ds.ids)
    n0, ni = eng.m_step(th0, thi, common_hp)
    return _hp_dict(n0, ni, eng.ds.ids)


def training(db, prior_mean, ini_hp, kern_0, kern_i, common_hp=True, trace=None, engine=None):
    """MG:15-82. Returns hp, convergence, param (last E-step)."""
    _check_kern(kern_0, kern_i)
    import time
    t1 = time.time()
    if engine is None:
        # the whole loop inside the library: ONE call (magma_training) per fit; in a
        # torch.distributed job every rank calls this with the same db and the library's callback
        # communicator sums over that group
        engine = engine_for_job(Dataset(_as_db(db)))
    eng = engine
    th0, thi, cv = eng.training(prior_mean, ini_hp["theta_0"], ini_hp["theta_i"], common_hp,
                                trace=trace)
    return {"hp": _hp_dict(th0, thi, eng.ds.ids), "convergence": cv,
            "param": _param_from_engine(eng), "Time_train": time.time() - t1,
            "stats": dict(eng.stats)}


def train_new_gp(db, mean_mu, cov_mu, ini_hp_i, kern_i):
    """MG:84-104."""
    _check_kern(kern_i)
    t = np.asarray(db["Timestamp"], dtype=np.float64)
    if isinstance(mean_mu, dict):
        # MG:86-88 `filter(Timestamp %in% db$Timestamp)`: rows in the mean tibble's own (sorted)
        # order, which is db order only for a db sorted by timestamp (the reference's assumption,
        # CF:74). Taken in db order here, like new_cov below, so that _indiv_problem permutes
        # mean, cov and outputs consistently.
        mt = np.asarray(mean_mu["Timestamp"], float)
        mo = np.asarray(mean_mu["Output"], float)
        srt = np.argsort(mt, kind="stable")
        pos = np.minimum(np.searchsorted(mt[srt], t), mt.size - 1)
        if np.any(mt[srt][pos] != t):
            raise KeyError("mean_mu does not cover db$Timestamp")
        mean = mo[srt][pos]
    else:
        mean = np.asarray(mean_mu, float).ravel()
    if isinstance(cov_mu, tuple):
        labels, cm = cov_mu
        pos = np.searchsorted(labels, t)
        if np.any(labels[np.minimum(pos, labels.size - 1)] != t):
            raise KeyError("subscript out of bounds")
        new_cov = cm[np.ix_(pos, pos)]
    else:
        new_cov = 0.0
    from .native import train_new_gp as native_train_new_gp
    order = np.argsort(t, kind="stable")
    y = np.asarray(db["Output"], dtype=np.float64)[order]
    mean = np.full(t.size, mean[0]) if mean.size == 1 else mean[order]
    cov = None if np.ndim(new_cov) == 0 else np.ascontiguousarray(np.asarray(new_cov, float)[np.ix_(order, order)])
    # one native call: the optimiser and MG:97-101's NaN rule run inside the library
    new_hp, _ = native_train_new_gp(t[order], y, mean, cov, np.asarray(ini_hp_i, dtype=np.float64))
    return {"theta": new_hp[:2], "sigma": new_hp[2]}


def posterior_mu(db, new_db, timestamps, m_0, kern_0, kern_i, hp, pen_diag=0.1):
    """MG:107-133. `pen_diag` is exposed (default: the committed 0.1, MG:116; 'mean_sigma'
    selects the commented-out variant that produced three of the four stored result files)."""
    _check_kern(kern_0, kern_i)
    extra = np.concatenate([np.asarray(timestamps, float).ravel(),
                            np.asarray(new_db["Timestamp"], float).ravel()])
    eng = engine_for_job(Dataset(_as_db(db), extra_timestamps=extra))
    th0, thi = _hp_arrays(hp, eng.ds.ids)
    if isinstance(pen_diag, str):
        pen_diag = float(np.mean(thi[:, 2]))
    info = eng.e_step(th0, thi, pen_diag, m_0)
    if info != 0:
        raise np.linalg.LinAlgError(f"hyper-posterior precision not positive definite ({info})")
    return _param_from_engine(eng)


def pred_gp(db, timestamps=None, mean_mu=0.0, cov_mu=None, kern=kernel, theta=(1.0, 0.2),
            sigma=0.2):
    """MG:135-181."""
    _check_kern(kern)
    import torch
    tn = np.asarray(db["Timestamp"], dtype=np.float64)
    yn = np.asarray(db["Output"], dtype=np.float64)
    if timestamps is None:
        timestamps = np.linspace(tn.min(), tn.max(), 500)
    ts = np.asarray(timestamps, dtype=np.float64).ravel()
    if isinstance(mean_mu, dict):
        mt, mo = np.asarray(mean_mu["Timestamp"], float), np.asarray(mean_mu["Output"], float)
        mean_obs = mo[np.isin(mt, tn)]
        mean_pred = mo[np.isin(mt, ts)]
    else:
        mean_obs = np.full(tn.size, float(np.ravel(mean_mu)[0]))
        mean_pred = np.full(ts.size, float(np.ravel(mean_mu)[0]))
    if mean_obs.size != tn.size or mean_pred.size != ts.size:
        raise ValueError("mean_mu does not cover the observation / prediction timestamps")
    dev = "cuda"
    f64 = torch.float64

    def up(a, dt=f64):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    N = 0
    d_cov = d_in = d_it = None
    if cov_mu is not None:
        labels, cm = cov_mu
        N = labels.size
        i_n = np.searchsorted(labels, tn)
        i_t = np.searchsorted(labels, ts)
        if np.any(labels[np.minimum(i_n, N - 1)] != tn) or np.any(labels[np.minimum(i_t, N - 1)] != ts):
            raise KeyError("subscript out of bounds")
        Np = _lib.load().magma_pad(N)
        c = np.eye(Np)
        c[:N, :N] = cm
        d_cov = up(c.T.reshape(-1))
        d_in = up(i_n.astype(np.int32), torch.int32)
        d_it = up(i_t.astype(np.int32), torch.int32)
    th3 = up(np.array([theta[0], theta[1], sigma], dtype=np.float64))
    d_tn, d_yn, d_mo = up(tn), up(yn), up(mean_obs)
    d_ts, d_mp = up(ts), up(mean_pred)
    o_mean = torch.empty(ts.size, dtype=f64, device=dev)
    o_var = torch.empty(ts.size, dtype=f64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("magma_pred_gp", tn.size, _ptr(d_tn), _ptr(d_yn), _ptr(d_mo), _ptr(d_in), ts.size,
              _ptr(d_ts), _ptr(d_mp), _ptr(d_it), _ptr(d_cov), N, _ptr(th3), _ptr(o_mean),
              _ptr(o_var), _ptr(info), st)
    # (info != 0: cov_tn_tn was answered by the library's solve() / ginv() slow path, MG:173)
    return {"Timestamp": ts, "Mean": o_mean.cpu().numpy(), "Var": o_var.cpu().numpy()}


def posterior_mu_resident(db, new_dbs, timestamps, m_0, kern_0, kern_i, hp, pen_diag=0.1):
    """posterior_mu (MG:107-133) on the union of `timestamps` and the timestamps of EVERY new
    individual in `new_dbs`, left on the device: returns the NativeEngine holding mean / cov in HBM for
    `pred_gp_batched`. The hyper-posterior at a timestamp does not depend on which
    other timestamps the grid holds (a GP posterior is consistent under marginalisation), so this
    equals the reference's per-individual posterior_mu up to rounding."""
    _check_kern(kern_0, kern_i)
    extra = [np.asarray(timestamps, float).ravel()]
    extra += [np.asarray(nd["Timestamp"], float).ravel() for nd in new_dbs]
    eng = engine_for_job(Dataset(_as_db(db), extra_timestamps=np.concatenate(extra)))
    th0, thi = _hp_arrays(hp, eng.ds.ids)
    if isinstance(pen_diag, str):
        pen_diag = float(np.mean(thi[:, 2]))
    info = eng.e_step(th0, thi, pen_diag, m_0)
    if info != 0:
        raise np.linalg.LinAlgError(f"hyper-posterior precision not positive definite ({info})")
    return eng


def pred_gp_batched(eng, new_dbs, timestamps, theta, sigma, kern=kernel):
    """pred_gp (MG:135-181) for many new individuals against the hyper-posterior resident in
    `eng` (from posterior_mu_resident): one launch, cov never leaves the device. `timestamps`:
    one array for all or a list; `theta` [B, 2] or [2], `sigma` [B] or scalar. Returns a list of
    {Timestamp, Mean, Var} like pred_gp."""
    _check_kern(kern)
    B = len(new_dbs)
    shared = not isinstance(timestamps, (list, tuple))
    tss = [np.asarray(timestamps if shared else timestamps[b], float).ravel() for b in range(B)]
    tn, yn, off_n, off_t = [], [], [0], [0]
    for b, nd in enumerate(new_dbs):
        t = np.asarray(nd["Timestamp"], dtype=np.float64)
        o = np.argsort(t, kind="stable")
        tn.append(t[o])
        yn.append(np.asarray(nd["Output"], dtype=np.float64)[o])
        off_n.append(off_n[-1] + t.size)
        off_t.append(off_t[-1] + tss[b].size)
    th = np.empty((B, 3))
    th[:, :2] = np.asarray(theta, dtype=np.float64).reshape(-1, 2)
    th[:, 2] = np.asarray(sigma, dtype=np.float64).ravel()
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0)  # noqa: E731
    # (info[b] != 0: individual b's cov_tn_tn went through solve() / ginv(), MG:173)
    mean, var, _ = eng.be.pred_gp_batched(off_n, cat(tn), cat(yn), off_t, cat(tss), th)
    return [{"Timestamp": tss[b], "Mean": mean[off_t[b]:off_t[b + 1]],
             "Var": var[off_t[b]:off_t[b + 1]]} for b in range(B)]


def full_algo(db, new_db, timestamps, kern_i, common_hp=True, plot=False, prior_mean=0.0,
              kern_0=None, list_hp=None, mu=None, ini_hp=None, hp_new_i=None, pen_diag=0.1):
    """MG:203-242 (plotting is out of scope: `plot` is accepted and ignored)."""
    if list_hp is None:
        list_hp = training(db, prior_mean, ini_hp, kern_0, kern_i, common_hp)["hp"]
    if mu is None:
        mu = posterior_mu(db, new_db, timestamps, prior_mean, kern_0, kern_i, list_hp, pen_diag)
    if hp_new_i is None and common_hp:
        first = next(iter(list_hp["theta_i"].values()))
        hp_new_i = {"theta": first[:2], "sigma": first[2]}
    elif hp_new_i is None:
        hp_new_i = train_new_gp(new_db, mu["mean"], mu["cov"], ini_hp["theta_i"], kern_i)
    pred = pred_gp(new_db, timestamps, mu["mean"], mu["cov"], kern_i, hp_new_i["theta"],
                   hp_new_i["sigma"])
    return {"Prediction": pred, "Mean_process": mu,
            "Hyperparameters": {"other_i": list_hp, "new_i": hp_new_i}}
