"""Host-side data layout and the kernel-level bindings.

`Dataset`    host-side ragged layout of a db (replaces the reference's per-call
             `filter(ID == i)` scans and `paste0('X', t)` name matching with offsets and an
             int32 union-grid index).
`HipBackend` one shard resident in HBM (torch tensors are only the allocator) and thin wrappers
             over the KERNEL-LEVEL entry points of libmagma_hip.so (magma_estep_*, magma_indiv_*,
             magma_theta0_*, magma_pred_gp_*): what the single-function API of `magma_amd.api`
             and the kernel parity tests call.

The EM logic itself (e_step / m_step / logL_monitoring / training, MAGMA.R:15-82,
Computing_functions.R:371-437, and the multi-rank reductions) lives in ONE place: the native
driver inside the library (csrc/driver.hip), reached through `magma_amd.native.NativeEngine`. The
Python mirror of that logic that earlier rounds kept next to it is test infrastructure now
(tests/host_mirror.py: it drives the oracle under gloo on CPU and cross-checks the driver).
"""
import ctypes as C
import numpy as np

from . import _lib

N_SMALL_MAX = 64


def unique_in_order(a):
    _, first = np.unique(a, return_index=True)
    return np.asarray(a)[np.sort(first)]


class Dataset:
    """Ragged individuals + sorted union grid. Each individual's (t, y) are sorted jointly by
    timestamp: the reference sorts timestamps inside kern_to_cov/kern_to_inv (CF:74,95,129,151)
    but takes outputs in db order, so it is only self-consistent for sorted input."""

    def __init__(self, db, extra_timestamps=None):
        ids_col = np.asarray(db["ID"])
        t_col = np.asarray(db["Timestamp"], dtype=np.float64)
        y_col = np.asarray(db["Output"], dtype=np.float64)
        # One grouping pass, O(R log R) in the number of rows: individuals in order
        # of first appearance (`unique(db$ID)`), rows of an individual by timestamp. Replaces a
        # `filter(ID == i)` scan per individual, which is O(M R).
        if ids_col.size:
            uniq, first, inv = np.unique(ids_col, return_index=True, return_inverse=True)
            rank = np.empty(uniq.size, dtype=np.int64)
            rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)
            grp = rank[inv.ravel()]
            order = np.lexsort((t_col, grp))  # stable: ties keep db order
            self.ids = list(uniq[np.argsort(first, kind="stable")])
            counts = np.bincount(grp, minlength=uniq.size)
            self.t = np.ascontiguousarray(t_col[order])
            self.y = np.ascontiguousarray(y_col[order])
            g_sorted = grp[order]
        else:
            self.ids, counts = [], np.zeros(0, dtype=np.int64)
            self.t, self.y, g_sorted = np.zeros(0), np.zeros(0), np.zeros(0, dtype=np.int64)
        self.M = len(self.ids)
        self.off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        pieces = [self.t]
        if extra_timestamps is not None:
            pieces.append(np.asarray(extra_timestamps, dtype=np.float64).ravel())
        self.grid = np.unique(np.concatenate(pieces))
        self.N = int(self.grid.size)
        self.idx = np.searchsorted(self.grid, self.t).astype(np.int32)
        # (individuals with more than 64 observations take the dense route of the native driver,
        # magma_amd.native; the kernel-level HipBackend handles one wavefront per individual)
        self.nmax = int(counts.max()) if counts.size else 1
        dup = (np.diff(self.t) <= 0) & (g_sorted[1:] == g_sorted[:-1])
        if np.any(dup):
            bad = self.ids[int(g_sorted[1:][dup][0])]
            raise ValueError(f"individual {bad!r} has duplicated timestamps")

    @classmethod
    def from_arrays(cls, off, t, y, grid=None, ids=None):
        """Build directly from ragged arrays (timestamps ascending per individual)."""
        self = cls.__new__(cls)
        self.off = np.asarray(off, dtype=np.int64)
        self.M = self.off.size - 1
        self.ids = list(range(self.M)) if ids is None else list(ids)
        self.t = np.ascontiguousarray(t, dtype=np.float64)
        self.y = np.ascontiguousarray(y, dtype=np.float64)
        self.grid = np.unique(self.t) if grid is None else np.asarray(grid, dtype=np.float64)
        self.N = int(self.grid.size)
        self.idx = np.searchsorted(self.grid, self.t).astype(np.int32)
        self.nmax = int(np.diff(self.off).max()) if self.M else 1
        return self

    def shard(self, rank, world):
        """Contiguous block of individuals for `rank` (equal counts: n_i is near-constant in
        the target workloads). Returns (lo, hi)."""
        per = (self.M + world - 1) // world
        lo = min(self.M, rank * per)
        return lo, min(self.M, lo + per)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class HipBackend:
    """One rank's shard of the dataset in HBM + the dense workspace, calling libmagma_hip.so."""

    def __init__(self, ds, lo, hi, device=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise _lib.MagmaHipError("no GPU visible: the HIP path has no CPU fallback")
        self.lib = _lib.load()
        if ds.nmax > N_SMALL_MAX:
            raise ValueError(f"an individual has {ds.nmax} observations; the kernel-level backend "
                             f"handles n_i <= {N_SMALL_MAX} (one wavefront per individual): use "
                             "magma_amd.native.NativeEngine, whose driver sends larger individuals "
                             "down the dense route")
        self.dev = torch.device(device if device is not None else "cuda")
        if self.dev.index is None:
            self.dev = torch.device("cuda", torch.cuda.current_device())
        self.N = ds.N
        self.Np = int(self.lib.magma_pad(ds.N))
        self.M = hi - lo
        self.nmax = ds.nmax
        f64, dev = torch.float64, self.dev
        o0, o1 = int(ds.off[lo]), int(ds.off[hi])
        self.n_i = np.diff(ds.off[lo:hi + 1])
        self._h_off = (ds.off[lo:hi + 1] - o0).astype(np.int64)
        self._h_idx = ds.idx[o0:o1].copy()
        # E-step sums: "gather" = fixed summation order (bitwise repeatable E-step, lists built
        # once per dataset), "atomic" = FP64 atomics straight from the inverting wavefronts
        import os
        self.scatter_route = os.environ.get("MAGMA_SCATTER", "gather")
        self._gl = None
        self.off = torch.from_numpy(ds.off[lo:hi + 1] - o0).to(dev)
        self.t = torch.from_numpy(ds.t[o0:o1].copy()).to(dev)
        self.y = torch.from_numpy(ds.y[o0:o1].copy()).to(dev)
        self.idx = torch.from_numpy(ds.idx[o0:o1].copy()).to(dev)
        gridp = np.zeros(self.Np)
        gridp[:ds.N] = ds.grid
        self.grid = torch.from_numpy(gridp).to(dev)
        Mb = max(self.M, 1)
        self.theta = torch.zeros(Mb * 3, dtype=f64, device=dev)
        self.f = torch.zeros(Mb, dtype=f64, device=dev)
        self.g = torch.zeros(Mb * 3, dtype=f64, device=dev)
        self.info = torch.zeros(Mb, dtype=torch.int32, device=dev)
        self.sel = torch.zeros(Mb, dtype=torch.int32, device=dev)
        self.P = torch.zeros(self.Np * self.Np, dtype=f64, device=dev)  # precision, then cov
        self.w = torch.zeros(self.Np, dtype=f64, device=dev)
        self.mean = torch.zeros(self.Np, dtype=f64, device=dev)
        self.m0 = torch.zeros(self.Np, dtype=f64, device=dev)
        wbytes = int(self.lib.magma_workspace_bytes(ds.N))
        self.work = torch.empty((wbytes + 7) // 8, dtype=f64, device=dev)
        self.cov = self.P  # alias: after estep_finish P holds the covariance
        self.cpack = torch.zeros(Mb * int(self.lib.magma_cpack_stride(self.nmax)), dtype=f64,
                                 device=dev)
        self._cpack_ok = False  # cpack holds cov[idx_i, idx_i] of the current cov
        self.n_evals = 0  # individual objective evaluations launched (for roofline figures)
        self._side = None
        self._a0_key = None  # (theta1, theta2, pen) whose K_0^-1 sits in the workspace
        pin = lambda n, dt: torch.empty(n, dtype=dt).pin_memory()  # noqa: E731
        self.h_theta, self.h_sel = pin(Mb * 3, f64), pin(Mb, torch.int32)
        self.h_f, self.h_g = pin(Mb, f64), pin(Mb * 3, f64)

    # -- plumbing
    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def sync(self):
        self.torch.cuda.current_stream(self.dev).synchronize()

    def compute_units(self):
        return int(self.torch.cuda.get_device_properties(self.dev).multi_processor_count)

    def run_aside(self, fn):
        """Run fn() in a helper thread whose HIP work goes to a side stream; returns a join
        function giving fn's result. The caller's stream must already be synchronised with
        the data fn reads (estep_finish synchronises)."""
        import threading
        torch = self.torch
        if self._side is None:
            # high priority: the theta_0 chain is a long sequence of small dependent kernels that
            # should not queue behind the wide per-individual launches it overlaps with
            self._side = torch.cuda.Stream(device=self.dev, priority=-1)
        box = {}

        def work():
            try:
                torch.cuda.set_device(self.dev)
                with torch.cuda.stream(self._side):
                    box["r"] = fn()
                    self._side.synchronize()
            except BaseException as e:  # re-raised by the joiner
                box["e"] = e
        th = threading.Thread(target=work)
        th.start()

        def join():
            th.join()
            if "e" in box:
                raise box["e"]
            return box["r"]
        return join

    def _set_theta(self, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        if th.ndim == 1:  # common hyper-parameters: one triple, stride 0
            self.theta[:3].copy_(self.torch.from_numpy(th))
            return 0
        assert th.shape == (self.M, 3), th.shape
        self.theta[:3 * self.M].copy_(self.torch.from_numpy(th.reshape(-1)))
        return 3

    def set_m0(self, m0):
        v = np.zeros(self.Np)
        m0 = np.asarray(m0, dtype=np.float64).ravel()
        v[:self.N] = m0[0] if m0.size == 1 else m0
        self.m0.copy_(self.torch.from_numpy(v))

    def set_mean_cov(self, mean, cov):
        """Load a hyper-posterior (host arrays on this engine's grid) into device buffers."""
        v = np.zeros(self.Np)
        v[:self.N] = mean
        self.mean.copy_(self.torch.from_numpy(v))
        c = np.eye(self.Np)
        c[:self.N, :self.N] = cov
        self.P.copy_(self.torch.from_numpy(c.T.reshape(-1).copy()))
        self._cpack_ok = False

    # -- E-step
    def estep_scatter(self, theta):
        """Zero P, w and scatter this shard's K_i^-1 / K_i^-1 y_i. Returns (P, w) tensors for
        the caller to all-reduce, and the number of flagged (non-SPD) individuals."""
        stride = self._set_theta(theta)
        self._cpack_ok = False
        self.P.zero_()
        self.w.zero_()
        if self.M > 0 and self.scatter_route == "gather":
            g = self._gather_lists()
            _lib.call("magma_estep_gather", self.M, self.nmax, _ptr(self.off), _ptr(g["off2"]),
                      _ptr(self.t), _ptr(self.y), _ptr(self.theta), stride, _ptr(g["Kinv"]),
                      _ptr(g["v"]), _ptr(self.f), g["nP"], _ptr(g["ptrP"]), _ptr(g["srcP"]),
                      _ptr(g["tgtP"]), g["nw"], _ptr(g["ptrw"]), _ptr(g["srcw"]), _ptr(g["tgtw"]),
                      _ptr(self.P), _ptr(self.w), _ptr(self.info), self.stream())
        elif self.M > 0:
            _lib.call("magma_estep_scatter", self.M, self.nmax, _ptr(self.off), _ptr(self.t),
                      _ptr(self.y), _ptr(self.idx), _ptr(self.theta), stride, self.N,
                      _ptr(self.P), _ptr(self.w), _ptr(self.info), self.stream())
        return self.P, self.w

    def _gather_lists(self):
        """Source lists of the deterministic E-step (magma_estep_gather), built once per dataset:
        for every element of the lower triangle of P (every entry of w) that receives anything,
        the positions of its terms in the materialised K_i^-1 (in v = K_i^-1 y_i), in a fixed
        order. Replaces the reference's `intersect(row.names)` block updates (CF:440-478)."""
        if self._gl is not None:
            return self._gl
        torch = self.torch
        off, idx, n = self._h_off, self._h_idx.astype(np.int64), self.n_i.astype(np.int64)
        off2 = np.concatenate([[0], np.cumsum(n * n)]).astype(np.int64)
        if off2[-1] >= 2 ** 31:
            raise ValueError("deterministic E-step: more than 2^31 matrix entries on one rank "
                             "(set MAGMA_SCATTER=atomic or use more ranks)")
        tg, sr = [], []
        for nn in np.unique(n):
            ids = np.nonzero(n == nn)[0]
            r, c = np.tril_indices(int(nn))  # r >= c: idx ascending, so the lower triangle of P
            for a in range(0, ids.size, 8192):  # bounded host temporaries
                blk = ids[a:a + 8192]
                o = off[blk][:, None]
                tg.append((idx[o + r[None, :]] + idx[o + c[None, :]] * self.Np).ravel())
                sr.append((off2[blk][:, None] + c[None, :] * nn + r[None, :]).ravel())
        dev = self.dev

        def csr(targets, sources):
            t = torch.from_numpy(targets).to(dev)
            ts, perm = torch.sort(t, stable=True)
            src = torch.from_numpy(sources).to(dev)[perm].to(torch.int32)
            uniq, counts = torch.unique_consecutive(ts, return_counts=True)
            ptr = torch.zeros(uniq.numel() + 1, dtype=torch.int64, device=dev)
            ptr[1:] = torch.cumsum(counts, 0)
            return int(uniq.numel()), ptr, src.contiguous(), uniq.contiguous()
        nP, ptrP, srcP, tgtP = csr(np.concatenate(tg), np.concatenate(sr))
        nw, ptrw, srcw, tgtw = csr(idx, np.arange(idx.size, dtype=np.int64))
        f64 = torch.float64
        self._gl = {"off2": torch.from_numpy(off2).to(dev), "nP": nP, "ptrP": ptrP, "srcP": srcP,
                    "tgtP": tgtP, "nw": nw, "ptrw": ptrw, "srcw": srcw, "tgtw": tgtw,
                    "Kinv": torch.empty(max(int(off2[-1]), 1), dtype=f64, device=dev),
                    "v": torch.empty(max(int(off[-1]), 1), dtype=f64, device=dev)}
        return self._gl

    def n_flagged(self):
        if self.M == 0:
            return 0
        return int((self.info[:self.M] != 0).sum().item())

    def estep_finish(self, theta0, pen_diag, want_f0=False):
        """Returns (info, f0): f0 = logL_GP_mod(theta_0) at the new hyper-posterior with this
        pen_diag (the theta_0 term of the next logL_monitoring), or None."""
        info = C.c_int(0)
        f0 = C.c_double(0.0)
        key = (float(theta0[0]), float(theta0[1]), float(pen_diag))
        reuse = int(self._a0_key == key)  # K_0^-1 left by the last f-only theta_0 evaluation
        _lib.call("magma_estep_finish", self.N, _ptr(self.grid), key[0], key[1], key[2],
                  _ptr(self.m0), _ptr(self.P), _ptr(self.w), _ptr(self.mean), _ptr(self.work),
                  reuse, C.byref(f0) if want_f0 else None, C.byref(info), self.stream())
        self._a0_key = key if info.value == 0 else None
        return info.value, (f0.value if want_f0 else None)

    # -- M-step objective / gradient
    def indiv_eval(self, theta, want_grad=True, mode=0, active=None, use_cov=True):
        """Per-individual f (and gradient) at `theta` ([M,3] or common [3]) against the device
        resident mean/cov. Returns host arrays f[M], g[M,3] (g None without want_grad)."""
        if self.M == 0:
            return np.zeros(0), (np.zeros((0, 3)) if want_grad else None)
        theta = np.asarray(theta, dtype=np.float64)
        sel, nsel, ids = None, 0, None
        if active is not None and theta.ndim == 2:
            ids = np.nonzero(active)[0].astype(np.int32)
            nsel = int(ids.size)
            if nsel == 0:
                return np.zeros(self.M), (np.zeros((self.M, 3)) if want_grad else None)
            if nsel < self.M:
                self.h_sel[:nsel].numpy()[:] = ids
                self.sel[:nsel].copy_(self.h_sel[:nsel], non_blocking=True)
                sel = self.sel
            else:
                ids = None
        torch = self.torch
        cnt = nsel if sel is not None else self.M
        if theta.ndim == 1:
            stride = self._set_theta(theta)
        else:  # one pinned staging copy per array, all asynchronous on the launch stream
            rows = theta[ids] if sel is not None else theta
            self.h_theta[:3 * cnt].numpy()[:] = np.ascontiguousarray(rows).reshape(-1)
            self.theta[:3 * cnt].copy_(self.h_theta[:3 * cnt], non_blocking=True)
            stride = 3
        if want_grad:
            self.n_evals += cnt
        if use_cov and not self._cpack_ok:  # once per hyper-posterior, not per evaluation
            _lib.call("magma_indiv_gather_cov", self.M, self.nmax, _ptr(self.off), _ptr(self.idx),
                      _ptr(self.cov), self.N, _ptr(self.cpack), self.stream())
            self._cpack_ok = True
        _lib.call("magma_indiv_eval", int(mode), self.M, self.nmax, _ptr(self.off), _ptr(self.t),
                  _ptr(self.y), _ptr(self.idx), _ptr(self.theta), stride, _ptr(self.mean),
                  _ptr(self.cpack if use_cov else None), int(bool(want_grad)), _ptr(sel),
                  nsel, _ptr(self.f), _ptr(self.g), _ptr(self.info), self.stream())
        self.h_f[:cnt].copy_(self.f[:cnt], non_blocking=True)
        if want_grad:
            self.h_g[:3 * cnt].copy_(self.g[:3 * cnt], non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()
        fc = self.h_f[:cnt].numpy().copy()
        gc = self.h_g[:3 * cnt].numpy().reshape(cnt, 3).copy() if want_grad else None
        if sel is None:
            return fc, gc
        f = np.zeros(self.M)
        f[ids] = fc
        g = None
        if want_grad:
            g = np.zeros((self.M, 3))
            g[ids] = gc
        return f, g

    def indiv_mstep(self, theta_start, mode=0, use_cov=True, m=5, factr=1e7, pgtol=0.0,
                    maxit=100):
        """All of this shard's per-individual L-BFGS-B problems, loop and all, in one native
        call (magma_indiv_mstep). Returns the dict minimize_batch returns."""
        M = self.M
        if M == 0:
            z = np.zeros(0)
            return {"x": np.zeros((0, 3)), "f": z, "f_x0": z, "iters": z.astype(np.int32),
                    "nfev": z.astype(np.int32), "status": z.astype(np.int32), "rounds": 0}
        x0 = np.ascontiguousarray(theta_start, dtype=np.float64)
        assert x0.shape == (M, 3), x0.shape
        if use_cov and not self._cpack_ok:
            _lib.call("magma_indiv_gather_cov", M, self.nmax, _ptr(self.off), _ptr(self.idx),
                      _ptr(self.cov), self.N, _ptr(self.cpack), self.stream())
            self._cpack_ok = True
        x = np.empty((M, 3))
        f, fx0 = np.empty(M), np.empty(M)
        iters, nfev, status = (np.empty(M, dtype=np.int32) for _ in range(3))
        rounds, nev = C.c_int(0), C.c_int64(0)
        hp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        _lib.call("magma_indiv_mstep", int(mode), M, self.nmax, _ptr(self.off), _ptr(self.t),
                  _ptr(self.y), _ptr(self.idx), _ptr(self.mean),
                  _ptr(self.cpack if use_cov else None), hp(x0), int(m), float(factr),
                  float(pgtol), int(maxit), _ptr(self.theta), _ptr(self.sel), _ptr(self.f),
                  _ptr(self.g), _ptr(self.info), hp(x), hp(f), hp(fx0), hp(iters), hp(nfev),
                  hp(status), C.byref(rounds), C.byref(nev), self.stream())
        self.n_evals += int(nev.value)
        return {"x": x, "f": f, "f_x0": fx0, "iters": iters, "nfev": nfev, "status": status,
                "rounds": int(rounds.value)}

    def indiv_mstep_resident(self, theta_start, mode=0, use_cov=True, m=5, factr=1e7, pgtol=0.0,
                             maxit=100):
        """The same problems solved by magma_indiv_mstep_resident: one launch, the optimiser
        runs inside the evaluating wavefronts."""
        M, torch = self.M, self.torch
        if M == 0:
            return self.indiv_mstep(theta_start, mode, use_cov, m, factr, pgtol, maxit)
        x0 = np.ascontiguousarray(theta_start, dtype=np.float64)
        assert x0.shape == (M, 3), x0.shape
        if use_cov and not self._cpack_ok:
            _lib.call("magma_indiv_gather_cov", M, self.nmax, _ptr(self.off), _ptr(self.idx),
                      _ptr(self.cov), self.N, _ptr(self.cpack), self.stream())
            self._cpack_ok = True
        if getattr(self, "_res", None) is None:
            f64, i32, dev = torch.float64, torch.int32, self.dev
            self._res = {"x": torch.empty(3 * M, dtype=f64, device=dev),
                         "f": torch.empty(M, dtype=f64, device=dev),
                         "fx0": torch.empty(M, dtype=f64, device=dev),
                         "it": torch.empty(M, dtype=i32, device=dev),
                         "nf": torch.empty(M, dtype=i32, device=dev),
                         "st": torch.empty(M, dtype=i32, device=dev)}
        r = self._res
        self.h_theta[:3 * M].numpy()[:] = x0.reshape(-1)
        self.theta[:3 * M].copy_(self.h_theta[:3 * M], non_blocking=True)
        order = None
        if getattr(self, "_last_nfev", None) is not None and self._last_nfev.shape[0] == M:
            # longest chains of the previous call first (they tend to stay the longest)
            perm = np.argsort(-self._last_nfev, kind="stable").astype(np.int32)
            self.h_sel[:M].numpy()[:] = perm
            self.sel[:M].copy_(self.h_sel[:M], non_blocking=True)
            order = self.sel
        _lib.call("magma_indiv_mstep_resident", int(mode), M, self.nmax, _ptr(self.off),
                  _ptr(self.t), _ptr(self.y), _ptr(self.idx), _ptr(self.mean),
                  _ptr(self.cpack if use_cov else None), _ptr(self.theta), _ptr(order), int(m),
                  float(factr),
                  float(pgtol), int(maxit), _ptr(r["x"]), _ptr(r["f"]), _ptr(r["fx0"]),
                  _ptr(r["it"]), _ptr(r["nf"]), _ptr(r["st"]), self.stream())
        out = {"x": r["x"].cpu().numpy().reshape(M, 3), "f": r["f"].cpu().numpy(),
               "f_x0": r["fx0"].cpu().numpy(), "iters": r["it"].cpu().numpy(),
               "nfev": r["nf"].cpu().numpy(), "status": r["st"].cpu().numpy()}
        out["rounds"] = int(out["nfev"].max())
        self._last_nfev = out["nfev"].astype(np.int64)
        self.n_evals += int(out["nfev"].sum())
        return out

    def theta0_eval(self, theta0, pen_diag, want_grad=True):
        out = (C.c_double * 3)()
        info = C.c_int(0)
        _lib.call("magma_theta0_eval", self.N, _ptr(self.grid), float(theta0[0]),
                  float(theta0[1]), float(pen_diag), _ptr(self.mean), _ptr(self.m0),
                  _ptr(self.cov), int(bool(want_grad)), _ptr(self.work), out, C.byref(info),
                  self.stream())
        self._a0_key = ((float(theta0[0]), float(theta0[1]), float(pen_diag))
                        if info.value == 0 else None)
        f = out[0] if info.value == 0 else float("nan")
        return f, np.array([out[1], out[2]]), info.value

    def theta0_mstep(self, theta0, pen_diag, m=5, factr=1e7, pgtol=0.0, maxit=100):
        """m_step's theta_0 problem (CF:412-413) as one native call (magma_theta0_mstep): the
        optimiser loop around theta0_eval runs in the library. Returns minimize_batch's dict."""
        x0 = (C.c_double * 2)(float(theta0[0]), float(theta0[1]))
        x, f, fx0 = (C.c_double * 2)(), C.c_double(0.0), C.c_double(0.0)
        it, nf, st = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.call("magma_theta0_mstep", self.N, _ptr(self.grid), x0, float(pen_diag),
                  _ptr(self.mean), _ptr(self.m0), _ptr(self.cov), int(m), float(factr),
                  float(pgtol), int(maxit), _ptr(self.work), x, C.byref(f), C.byref(fx0),
                  C.byref(it), C.byref(nf), C.byref(st), self.stream())
        self._a0_key = None  # the workspace holds K_0^-1 of the last TRIAL point
        return {"x": np.array([[x[0], x[1]]]), "f": np.array([f.value]),
                "f_x0": np.array([fx0.value]), "iters": np.array([it.value], dtype=np.int32),
                "nfev": np.array([nf.value], dtype=np.int32),
                "status": np.array([st.value], dtype=np.int32)}

    def pred_gp_batched(self, off_n, tn, yn, off_t, ts, theta):
        """pred_gp (MG:172-180) for B new individuals against the hyper-posterior resident in
        this backend (mean / cov as estep_finish left them), one launch (magma_pred_gp_batched).
        Ragged host arrays; every timestamp must lie on this backend's grid. theta: [B, 3] or [3].
        Returns (mean, var, info[B]) as host arrays, ragged like ts."""
        torch = self.torch
        off_n = np.ascontiguousarray(off_n, dtype=np.int64)
        off_t = np.ascontiguousarray(off_t, dtype=np.int64)
        B = off_n.size - 1
        tn = np.ascontiguousarray(tn, dtype=np.float64)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        if B == 0:
            return np.zeros(0), np.zeros(0), np.zeros(0, dtype=np.int32)
        grid = self.grid[:self.N].cpu().numpy()

        def on_grid(x):
            pos = np.minimum(np.searchsorted(grid, x), self.N - 1)
            if np.any(grid[pos] != x):
                raise KeyError("subscript out of bounds")  # MG:161-176 index by name
            return pos.astype(np.int32)
        i_n, i_t = on_grid(tn), on_grid(ts)
        nmax, tmax = int(np.diff(off_n).max()), int(np.diff(off_t).max())
        if nmax > N_SMALL_MAX or int(np.diff(off_n).min()) < 1:
            raise ValueError("pred_gp_batched: 1 <= n_b <= 64 observations per individual")
        th = np.ascontiguousarray(theta, dtype=np.float64)
        stride = 0 if th.ndim == 1 else 3
        assert th.shape == ((3,) if stride == 0 else (B, 3)), th.shape
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)  # noqa: E731
        d = [up(a) for a in (off_n, tn, np.asarray(yn, dtype=np.float64), i_n, off_t, ts, i_t,
                             th.reshape(-1))]
        o_mean = torch.empty(max(ts.size, 1), dtype=torch.float64, device=self.dev)
        o_var = torch.empty(max(ts.size, 1), dtype=torch.float64, device=self.dev)
        info = torch.zeros(B, dtype=torch.int32, device=self.dev)
        _lib.call("magma_pred_gp_batched", B, nmax, tmax, _ptr(d[0]), _ptr(d[1]), _ptr(d[2]),
                  _ptr(d[3]), _ptr(d[4]), _ptr(d[5]), _ptr(d[6]), _ptr(self.mean), _ptr(self.cov),
                  self.N, _ptr(d[7]), stride, _ptr(o_mean), _ptr(o_var), _ptr(info),
                  self.stream())
        return (o_mean[:ts.size].cpu().numpy(), o_var[:ts.size].cpu().numpy(),
                info.cpu().numpy())

    # -- results back to the host
    def get_mean(self):
        return self.mean[:self.N].cpu().numpy()

    def get_cov(self):
        c = self.P.view(self.Np, self.Np)[:self.N, :self.N]
        return c.t().contiguous().cpu().numpy()  # column-major -> numpy [row, col]

    def get_cov_diag(self):
        return self.P.view(self.Np, self.Np).diagonal()[:self.N].cpu().numpy()
