"""Synthetic datasets of the reference's simulation recipe (`simu_scheme`,
simulation_study.R:88-124, with the intervals actually used for the stored data, :470-472):
m_0(t) = a t + b, mu_0 ~ GP(m_0, SE(theta_0)), y_i ~ N(mu_0(t_i), SE(theta_i) + sigma_i^2 I),
t_i = n sorted grid points drawn without replacement from G (N equally spaced points on [0, 10]).

RNG is NumPy's `default_rng(seed)` (R's stream is not reproducible outside R). The mean-process
draw uses random Fourier features (2048 features) instead of an N x N factorisation, so large
grids stay cheap; the individual draws use exact 2-decimals-rounded hyper-parameters and a
batched Cholesky with 1e-8 e^theta1 jitter (sigma may round to 0.00).
"""
import numpy as np


def simu_arrays(M, n, N, common_times=False, common_hp=False, seed=1001, chunk=4096):
    """Returns dict(off, t, y, grid, theta_true[M,3], theta0_true[2]) — ragged arrays ready for
    `Dataset.from_arrays`."""
    rng = np.random.default_rng(seed)
    G = np.linspace(0.0, 10.0, N)
    a = round(rng.uniform(-2, 2), 2)
    b = round(rng.uniform(0, 10), 2)
    m0 = a * G + b
    mu_a = round(rng.uniform(0, 5), 2)
    mu_b = round(rng.uniform(0, 2), 2)
    D = 2048
    ell = np.exp(mu_b / 2.0)  # k(d) = e^a exp(-d^2 / (2 e^b))
    om = rng.normal(0.0, 1.0 / ell, D)
    ph = rng.uniform(0, 2 * np.pi, D)
    mu0 = m0 + np.sqrt(2.0 * np.exp(mu_a) / D) * np.cos(G[:, None] * om[None, :] + ph[None, :]).sum(1)

    def draw_hp(k):
        return np.stack([np.round(rng.uniform(0, 5, k), 2), np.round(rng.uniform(0, 2, k), 2),
                         np.round(rng.uniform(0, 2, k), 2)], axis=1)
    theta = np.tile(draw_hp(1), (M, 1)) if common_hp else draw_hp(M)
    t = np.empty((M, n))
    y = np.empty((M, n))
    common_idx = np.sort(rng.choice(N, n, replace=False)) if common_times else None
    for s in range(0, M, chunk):
        e = min(M, s + chunk)
        k = e - s
        if common_times:
            idx = np.tile(common_idx, (k, 1))
        else:
            idx = np.sort(np.argsort(rng.random((k, N)), axis=1)[:, :n], axis=1)
        ti = G[idx]
        d2 = (ti[:, :, None] - ti[:, None, :]) ** 2
        th = theta[s:e]
        K = np.exp(th[:, 0, None, None] - 0.5 * np.exp(-th[:, 1, None, None]) * d2)
        diag = (th[:, 2] ** 2 + 1e-8 * np.exp(th[:, 0]))[:, None, None] * np.eye(n)[None]
        L = np.linalg.cholesky(K + diag)
        z = rng.standard_normal((k, n, 1))
        y[s:e] = mu0[idx] + (L @ z)[:, :, 0]
        t[s:e] = ti
    off = np.arange(M + 1, dtype=np.int64) * n
    return {"off": off, "t": t.reshape(-1), "y": y.reshape(-1), "grid": G, "theta_true": theta,
            "theta0_true": np.array([mu_a, mu_b]), "mu0": mu0}


def simu_db(M, n, N, **kw):
    """Same data as a db dict (ID, Timestamp, Output) for the R-named API."""
    a = simu_arrays(M, n, N, **kw)
    ids = np.repeat(np.arange(1, M + 1), n)
    return {"ID": ids, "Timestamp": a["t"], "Output": a["y"]}, a
