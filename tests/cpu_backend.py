"""Test-only backend: the same interface as magma_amd.engine.HipBackend, computed with the CPU
oracle. Lets the host logic of `Engine` (sharding, reductions over ranks, the EM driver, the
lock-step optimiser plumbing) run without a GPU, e.g. under gloo with world_size 2.
NOT part of the product: magma_amd never imports this."""
import numpy as np
import torch

from oracle import magma_oracle as orc


class OracleBackend:
    def __init__(self, ds, lo, hi):
        self.ds, self.lo, self.hi = ds, lo, hi
        self.M = hi - lo
        self.N = ds.N
        self.grid = ds.grid
        self.P = torch.zeros(ds.N * ds.N, dtype=torch.float64)
        self.w = torch.zeros(ds.N, dtype=torch.float64)
        self.m0 = np.zeros(ds.N)
        self.mean = np.zeros(ds.N)
        self.cov = np.eye(ds.N)
        self.n_evals = 0

    def _indiv(self, k):
        i = self.lo + k
        s = slice(self.ds.off[i], self.ds.off[i + 1])
        return self.ds.t[s], self.ds.y[s], self.ds.idx[s]

    def set_m0(self, m0):
        m0 = np.asarray(m0, float).ravel()
        self.m0 = np.full(self.N, m0[0]) if m0.size == 1 else m0.copy()

    def set_mean_cov(self, mean, cov):
        self.mean, self.cov = np.asarray(mean, float).copy(), np.asarray(cov, float).copy()

    def estep_scatter(self, theta):
        P = np.zeros((self.N, self.N))
        w = np.zeros(self.N)
        for k in range(self.M):
            t, y, idx = self._indiv(k)
            th = theta if np.ndim(theta) == 1 else theta[k]
            _, inv = orc.kern_to_inv(t, orc.kernel, th[:2], th[2])
            P[np.ix_(idx, idx)] += inv
            w[idx] += inv @ y
        self.P.copy_(torch.from_numpy(P.reshape(-1)))
        self.w.copy_(torch.from_numpy(w))
        return self.P, self.w

    def n_flagged(self):
        return 0

    def estep_finish(self, theta0, pen_diag, want_f0=False):
        _, inv0 = orc.kern_to_inv(self.grid, orc.kernel, theta0, pen_diag)
        P = self.P.numpy().reshape(self.N, self.N) + inv0
        w = self.w.numpy() + inv0 @ self.m0
        self.cov = np.linalg.inv(P)
        self.mean = self.cov @ w
        f0 = self.theta0_eval(theta0, pen_diag, False)[0] if want_f0 else None
        return 0, f0

    def indiv_eval(self, theta, want_grad=True, mode=0, active=None, use_cov=True):
        f = np.zeros(self.M)
        g = np.zeros((self.M, 3)) if want_grad else None
        for k in range(self.M):
            if active is not None and not active[k]:
                continue
            t, y, idx = self._indiv(k)
            th = np.asarray(theta if np.ndim(theta) == 1 else theta[k], float)
            db = {"Timestamp": t, "Output": y}
            c = self.cov[np.ix_(idx, idx)]
            fn, gr = (orc.logL_GP_mod, orc.gr_GP_mod) if mode == 0 else (orc.logL_GP, orc.gr_GP)
            f[k] = fn(th, db, self.mean[idx], orc.kernel, c)
            if want_grad:
                g[k] = gr(th, db, self.mean[idx], orc.kernel, c)
        return f, g

    def theta0_eval(self, theta0, pen_diag, want_grad=True):
        db = {"Timestamp": self.grid, "Output": self.mean}
        th = np.asarray(theta0, float)
        f = orc.logL_GP_mod(th, db, self.m0, orc.kernel, self.cov, pen_diag)
        g = orc.gr_GP_mod(th, db, self.m0, orc.kernel, self.cov, pen_diag) if want_grad else np.zeros(2)
        return f, g, 0

    def get_mean(self):
        return self.mean.copy()

    def get_cov(self):
        return self.cov.copy()
