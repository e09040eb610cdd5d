"""TEST INFRASTRUCTURE: a Python mirror of the native driver's EM logic (csrc/driver.hip) over a
pluggable backend. Not part of the product: `magma_amd` runs the EM through
`magma_amd.native.NativeEngine` only. Kept because it
  * drives tests/cpu_backend.OracleBackend under torch.distributed/gloo on CPU (world_size 2: the
    sharding + reduction pattern of SURVEY 8e without a GPU), and
  * cross-checks the driver on the GPU through the kernel-level bindings (HipBackend), step by step.
MAGMA.R:15-82 (training), Computing_functions.R:371-437 (e_step, m_step), :236-264 (logL_monitoring).
"""
import numpy as np

from magma_amd import _lib
from magma_amd.engine import HipBackend
from magma_amd.lbfgsb import minimize_batch


class Comm:
    """torch.distributed plumbing; `None` group = single process."""

    def __init__(self, dist=None):
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def allreduce_(self, tensor):
        if self.world > 1:
            self.dist.all_reduce(tensor)
        return tensor

    def allreduce_host(self, arr, like=None):
        """Sum a small host array over ranks (through a tensor on the backend's device)."""
        if self.world == 1:
            return arr
        import torch
        dev = like.device if like is not None else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(dev)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def bcast_host(self, arr, like=None, src=0):
        if self.world == 1:
            return arr
        import torch
        dev = like.device if like is not None else "cpu"
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(dev)
        self.dist.broadcast(t, src)
        return t.cpu().numpy()

    def allgather_rows(self, local, counts, like=None):
        """Concatenate per-rank [m_r, k] arrays in rank order (m_r = counts[r])."""
        if self.world == 1:
            return local
        import torch
        dev = like.device if like is not None else "cpu"
        k = local.shape[1]
        mx = max(counts)
        buf = np.zeros((mx, k))
        buf[:local.shape[0]] = local
        t = torch.from_numpy(buf).to(dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)


class Engine:
    """EM logic on a backend. All hyper-parameter arrays are host NumPy; theta_i is the global
    [M, 3] array replicated on every rank, each rank evaluates rows lo:hi."""

    def __init__(self, ds, backend_factory=None, comm=None, device=None):
        self.ds = ds
        self.comm = comm or Comm(None)
        self.lo, self.hi = ds.shard(self.comm.rank, self.comm.world)
        self.counts = [ds.shard(r, self.comm.world)[1] - ds.shard(r, self.comm.world)[0]
                       for r in range(self.comm.world)]
        factory = backend_factory or (lambda d, lo, hi: HipBackend(d, lo, hi, device))
        self.be = factory(ds, self.lo, self.hi)
        self.stats = {"theta0_evals": 0, "indiv_rounds": 0, "common_evals": 0, "flagged": 0}
        # theta_i problems of the M-step: "resident" = optimiser inside the evaluating wavefronts
        # (one launch); True = lock-step rounds driven by the library (magma_indiv_mstep);
        # False = lock-step rounds driven from Python (minimize_batch)
        import os
        self.native_mstep = {"resident": "resident", "native": True, "python": False}[
            os.environ.get("MAGMA_MSTEP", "resident")]
        # Compute units kept free of the resident per-individual kernel while the theta_0 chain
        # runs next to it (its workgroups fill whole compute units; the chain's dense inverse
        # needs whole free ones): the resident launch is capped at (CUs - reserve) workgroups and
        # the one-launch inverse at `reserve` for the duration of the M-step. 0 = no split.
        self.reserve_cus = int(os.environ.get("MAGMA_RESERVE_CUS", 96))
        self._dag_wgs = int(os.environ.get("MAGMA_DAG_WGS", 256))
        # workgroups of the one-launch dense inverse (library default 256; 0 = launch-per-step path)
        if os.environ.get("MAGMA_DAG_WGS") is not None:
            _lib.call("magma_set_option", b"dag_workgroups", int(os.environ["MAGMA_DAG_WGS"]))

    # ------------------------------------------------------------ E-step (CF:371-395)
    def e_step(self, theta0, theta_i, pen_diag, m0):
        be = self.be
        be.set_m0(m0)
        common = np.ndim(theta_i) == 1
        P, w = be.estep_scatter(theta_i if common else theta_i[self.lo:self.hi])
        self.comm.allreduce_(P)
        self.comm.allreduce_(w)
        info, self.last_f0 = be.estep_finish(theta0, pen_diag, want_f0=True)
        self.stats["flagged"] += be.n_flagged()  # (after the finish: its read-back synchronises)
        return info

    # ------------------------------------------------------------ objectives
    def f0(self, theta0, pen_diag, want_grad=True):
        self.stats["theta0_evals"] += 1
        f, g, _ = self.be.theta0_eval(theta0, pen_diag, want_grad)
        return f, g

    def sum_fi(self, theta_i, want_grad=False):
        """Sum over ALL individuals of logL_GP_mod (and gradient sum): the common-HP objective
        (CF:203-234, 334-368) or the individual part of logL_monitoring (CF:254-261)."""
        common = np.ndim(theta_i) == 1
        f, g = self.be.indiv_eval(theta_i if common else theta_i[self.lo:self.hi], want_grad)
        loc = np.zeros(4)
        loc[0] = f.sum()
        if want_grad:
            loc[1:] = g.sum(axis=0)
        tot = self.comm.allreduce_host(loc, like=getattr(self.be, "w", None))
        return tot[0], tot[1:]

    def logL_monitoring(self, theta0, theta_i):
        """CF:236-264."""
        sig = theta_i[2] if np.ndim(theta_i) == 1 else theta_i[:, 2].mean()
        ll_0, _ = self.f0(theta0, sig, want_grad=False)
        s, _ = self.sum_fi(theta_i, want_grad=False)
        return -ll_0 - s

    # ------------------------------------------------------------ M-step (CF:397-437)
    def m_step(self, theta0, theta_i, common_hp, opt_kw=None):
        kw = dict(m=5, factr=1e7, pgtol=0.0, maxit=100)
        kw.update(opt_kw or {})
        like = getattr(self.be, "w", None)
        pen_diag = float(np.mean(2.0 * theta_i[:, 2]))  # CF:409

        def ev0(x, act):
            f, g = self.f0(x[0], pen_diag, True)
            return np.array([f]), g[None, :]

        def run0():
            if hasattr(self.be, "theta0_mstep") and self.native_mstep:
                r = self.be.theta0_mstep(theta0, pen_diag, **kw)
                self.stats["theta0_evals"] += int(r["nfev"][0])
                return r
            return minimize_batch(np.asarray(theta0, dtype=np.float64)[None, :], ev0, **kw)
        # theta_0 (few dense N x N evaluations, small grids) and the theta_i problems (many
        # wide launches with host round trips) are independent given the E-step: run theta_0
        # on a side stream in a helper thread so that the two overlap on the GPU.
        aside = hasattr(self.be, "run_aside") and not common_hp
        split = (aside and self.reserve_cus > 0 and self.hi > self.lo
                 and self.native_mstep == "resident" and self._dag_wgs > 0)
        if split:
            ncu = self.be.compute_units()
            _lib.call("magma_set_option", b"resident_max_workgroups", max(1, ncu - self.reserve_cus))
        elif aside:
            _lib.call("magma_set_option", b"resident_max_workgroups", 0)
        fut0 = self.be.run_aside(run0) if aside else None
        if fut0 is None:
            r0 = run0()

        if common_hp:
            def evc(x, act):
                self.stats["common_evals"] += 1
                f, g = self.sum_fi(x[0], True)
                return np.array([f]), g[None, :]
            rc = minimize_batch(theta_i[0][None, :], evc, **kw)
            self.last_sum_fi = float(rc["f_x0"][0])  # already summed over ranks
            self.last_sum_fi_new = float(rc["f"][0])
            par = self.comm.bcast_host(rc["x"][0], like=like)
            new_theta_i = np.tile(par, (theta_i.shape[0], 1))
        else:
            def evi(x, act):
                self.stats["indiv_rounds"] += 1
                return self.be.indiv_eval(x, True, active=act)
            if self.hi > self.lo:
                if hasattr(self.be, "indiv_mstep_resident") and self.native_mstep == "resident":
                    ri = self.be.indiv_mstep_resident(theta_i[self.lo:self.hi], **kw)
                    self.stats["indiv_rounds"] += ri["rounds"]
                    redo = np.nonzero(ri["status"] == 6)[0]
                    if redo.size:  # flagged start points: the lock-step route (as magma_m_step does)
                        rl = self.be.indiv_mstep(theta_i[self.lo:self.hi], **kw)
                        for key in ("x", "f", "f_x0", "iters", "nfev", "status"):
                            ri[key][redo] = rl[key][redo]
                elif hasattr(self.be, "indiv_mstep") and self.native_mstep:
                    ri = self.be.indiv_mstep(theta_i[self.lo:self.hi], **kw)
                    self.stats["indiv_rounds"] += ri["rounds"]
                else:
                    ri = minimize_batch(theta_i[self.lo:self.hi], evi, **kw)
                loc, fsum, fnew = ri["x"], float(np.sum(ri["f_x0"])), float(np.sum(ri["f"]))
            else:
                loc, fsum, fnew = np.zeros((0, 3)), 0.0, 0.0
            tot = self.comm.allreduce_host(np.array([fsum, fnew]), like=like)
            self.last_sum_fi, self.last_sum_fi_new = float(tot[0]), float(tot[1])
            new_theta_i = self.comm.allgather_rows(loc, self.counts, like=like)
        if fut0 is not None:
            r0 = fut0()
        if split:
            _lib.call("magma_set_option", b"resident_max_workgroups", 0)
        new_theta0 = self.comm.bcast_host(r0["x"][0], like=like)
        return new_theta0, new_theta_i

    # ------------------------------------------------------------ EM driver (MG:15-82)
    def em_iteration(self, theta0, theta_i, m0, common_hp, opt_kw=None):
        """One pass of the loop body of `training` (MG:32-75) without the stopping logic:
        E-step, M-step, the two logL_monitoring calls. Returns (new_theta0, new_theta_i,
        logLL at hp, logLL at new hp)."""
        pen_e = float(np.mean(theta_i[:, 2]))  # CF:381 (signed mean)
        info = self.e_step(theta0, theta_i, pen_e, m0)
        if info != 0:
            # K_0 or the hyper-posterior precision is not positive definite (or the one-launch
            # inverse gave up, -999): mean / cov are garbage and nothing may be optimised on them
            # (the reference would go through MASS::ginv here, CF:133,387)
            raise np.linalg.LinAlgError(
                f"E-step: {'K_0' if info > 0 else 'hyper-posterior precision'} not positive "
                f"definite (info={info})")
        new_theta0, new_theta_i = self.m_step(theta0, theta_i, common_hp, opt_kw)
        if np.any(np.isnan(new_theta0)) or np.any(np.isnan(new_theta_i)):
            return new_theta0, new_theta_i, np.nan, np.nan
        # logL_monitoring(hp) (MG:54) costs nothing extra: its theta_0 term came out of the
        # E-step's K_0^-1 (same theta_0, same pen_diag = mean sigma_i: CF:251, 381) and its
        # individual terms are the optimisers' first evaluations at the old theta_i.
        l_old = -self.last_f0 - self.last_sum_fi
        # logL_monitoring(new hp) (MG:60): its individual terms are the optimisers' objective
        # values at the points they returned (same mean, cov); only the theta_0 term needs an
        # evaluation (pen_diag changes from mean 2 sigma_i to mean sigma_i of the new hp), and
        # that one leaves K_0^-1 behind for the next E-step.
        sig = float(new_theta_i[:, 2].mean())
        ll_0, _ = self.f0(new_theta0, sig, want_grad=False)
        l_new = -ll_0 - self.last_sum_fi_new
        return new_theta0, new_theta_i, l_old, l_new

    def training(self, prior_mean, ini_theta0, ini_theta_i, common_hp=True, n_loop_max=25,
                 trace=None, opt_kw=None):
        M = self.ds.M
        theta0 = np.asarray(ini_theta0, dtype=np.float64).copy()
        theta_i = np.tile(np.asarray(ini_theta_i, dtype=np.float64), (M, 1))
        cv = False
        for it in range(1, n_loop_max + 1):
            n0, ni, l_old, l_new = self.em_iteration(theta0, theta_i, prior_mean, common_hp,
                                                     opt_kw)
            if np.isnan(l_old):  # MG:46-51: keep the last valid iteration
                break
            eps = (l_new - l_old) / abs(l_new)
            if trace is not None:
                trace.append({"iter": it, "logLL": l_old, "logL_new_hp": l_new, "eps": eps})
            if eps < 1e-3:  # MG:66-71
                if eps > 0:
                    theta0, theta_i = n0, ni
                cv = True
                break
            theta0, theta_i = n0, ni
        return theta0, theta_i, cv
