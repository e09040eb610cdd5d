"""Thin host-side caller of the native EM driver (include/magma_hip.h, "context / dataset / driver":
magma_ctx / magma_dataset / magma_training ...). Everything the reference's `training`,
`e_step`, `m_step`, `logL_monitoring`, `posterior_mu`, `train_new_gp` do (MAGMA.R:15-133,
Computing_functions.R:236-264, 371-437) happens inside libmagma_hip.so; this module only turns
NumPy arrays into pointers. The multi-GPU communicator lives in the library too: RCCL from a
unique id, or an all-reduce callback (tests: several ranks as threads on one GPU).
"""
import ctypes as C
import numpy as np

from . import _lib

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class EmStats(C.Structure):
    _fields_ = [("theta0_evals", C.c_int64), ("indiv_rounds", C.c_int64), ("indiv_evals", C.c_int64),
                ("common_evals", C.c_int64), ("flagged", C.c_int64),
                ("slow_resolved", C.c_int64)]


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to every rank)."""
    buf = (C.c_char * 128)()
    _lib.call("magma_comm_unique_id", C.cast(buf, C.c_void_p))
    return bytes(buf)


def device_tensor(ptr, n):
    """A float64 torch view of `n` doubles of device memory at `ptr` (for all-reduce callbacks)."""
    import torch

    class _Dev:
        pass
    d = _Dev()
    d.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                  "version": 2}
    return torch.as_tensor(d, device="cuda")


def dist_allreduce(dist):
    """All-reduce callback for the library's callback communicator over an initialised
    torch.distributed group: RCCL ("nccl") sums the device buffer in place; any other backend
    (gloo: tests, rehearsals) stages it through the host."""
    import torch
    on_device = dist.get_backend() == "nccl"

    def allreduce(ptr, count):
        tens = device_tensor(ptr, count)
        if on_device:
            torch.cuda.synchronize()
            dist.all_reduce(tens)
        else:
            host = tens.cpu()
            dist.all_reduce(host)
            tens.copy_(host)
        torch.cuda.synchronize()
    return allreduce


def engine_for_job(ds, device=None):
    """This process's NativeEngine on `ds`: alone, or rank r of an initialised torch.distributed
    job (the library's callback communicator over that group; contiguous shards of individuals)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return NativeEngine(ds, rank=dist.get_rank(), world=dist.get_world_size(),
                                allreduce=dist_allreduce(dist), device=device)
    except ImportError:  # pragma: no cover
        pass
    return NativeEngine(ds, device=device)


class NativeEngine:
    """One rank of the native driver: a context (device, streams, communicator) and this rank's
    shard of `ds` (magma_amd.engine.Dataset) resident on the device.

    rank / world with `unique_id` -> RCCL inside the library; with `allreduce` (a Python function
    (device_ptr, count) -> None summing in place over ranks) -> the callback communicator."""

    def __init__(self, ds, rank=0, world=1, unique_id=None, allreduce=None, device=None):
        import torch
        if not torch.cuda.is_available():
            raise _lib.MagmaHipError("no GPU visible: the HIP path has no CPU fallback")
        self.lib = _lib.load()
        self.ds = ds
        self.rank, self.world = int(rank), int(world)
        dev = torch.device(device if device is not None else "cuda")
        self.device = dev.index if dev.index is not None else torch.cuda.current_device()
        per = (ds.M + world - 1) // world
        self.lo = min(ds.M, rank * per)
        self.hi = min(ds.M, self.lo + per)
        self._ctx = C.c_void_p(0)
        self._ds = C.c_void_p(0)
        self._cb = None
        if unique_id is not None:  # (also with world == 1: a one-rank RCCL communicator)
            _lib.call("magma_ctx_create_rccl", C.byref(self._ctx), self.device, self.rank, self.world,
                      C.c_char_p(unique_id))
        elif world > 1 or allreduce is not None:
            if allreduce is None:
                raise ValueError("world > 1 needs a unique_id (RCCL) or an allreduce callback")

            def cb(user, ptr, count, stream):
                try:
                    allreduce(ptr, count)
                    return 0
                except BaseException:  # noqa: BLE001  (reported by the library as status 210)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = ALLREDUCE_FN(cb)  # keep alive
            _lib.call("magma_ctx_create_callback", C.byref(self._ctx), self.device, self.rank,
                      self.world, C.cast(self._cb, C.c_void_p), None)
        else:
            _lib.call("magma_ctx_create", C.byref(self._ctx), self.device)
        o0, o1 = int(ds.off[self.lo]), int(ds.off[self.hi])
        off = np.ascontiguousarray(ds.off[self.lo:self.hi + 1] - o0, dtype=np.int64)
        t, y = _f64(ds.t[o0:o1]), _f64(ds.y[o0:o1])
        grid = _f64(ds.grid)
        _lib.call("magma_dataset_create", self._ctx, C.byref(self._ds), self.hi - self.lo, ds.M,
                  self.lo, _hp(off), _hp(t), _hp(y), ds.N, _hp(grid))
        self.N = ds.N
        self.last_f0 = None
        self.be = self  # (api._param_from_engine reads eng.be.get_mean / get_cov)

    def close(self):
        if self._ds:
            self.lib.magma_dataset_destroy(self._ds)
            self._ds = C.c_void_p(0)
        if self._ctx:
            self.lib.magma_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p(0)

    def selfcheck(self, timeout_ms=20000):
        """magma_ctx_selfcheck: a known vector through the communicator on both streams (collective);
        raises MagmaHipError (status 211: did not complete, 212: wrong sum)."""
        _lib.call("magma_ctx_selfcheck", self._ctx, int(timeout_ms))

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover  (interpreter shutdown)
            pass

    # ------------------------------------------------------------ helpers
    @staticmethod
    def _opt(opt_kw):
        kw = dict(m=5, factr=1e7, pgtol=0.0, maxit=100)
        kw.update(opt_kw or {})
        return np.array([kw["m"], kw["factr"], kw["pgtol"], kw["maxit"]], dtype=np.float64)

    def _theta_i(self, theta_i):
        th = _f64(theta_i)
        if th.ndim == 1:
            return th, 0
        assert th.shape == (self.ds.M, 3), th.shape
        return th, 3

    @staticmethod
    def _m0(m0):
        v = _f64(np.ravel(m0))
        return v, int(v.size)

    @property
    def stats(self):
        s = EmStats()
        _lib.call("magma_get_stats", self._ds, C.byref(s), 0)
        return {k: int(getattr(s, k)) for k, _ in EmStats._fields_}

    def reset_stats(self):
        s = EmStats()
        _lib.call("magma_get_stats", self._ds, C.byref(s), 1)

    # ------------------------------------------------------------ reference functions
    def e_step(self, theta0, theta_i, pen_diag, m0):
        th0 = _f64(theta0)
        th, stride = self._theta_i(theta_i)
        mv, nm = self._m0(m0)
        info, f0 = C.c_int(0), C.c_double(0.0)
        _lib.call("magma_e_step", self._ctx, self._ds, _hp(th0), _hp(th), stride, float(pen_diag),
                  _hp(mv), nm, C.byref(info), C.byref(f0))
        self.last_f0 = f0.value
        return info.value

    def m_step(self, theta0, theta_i, common_hp, opt_kw=None):
        th0, th = _f64(theta0), _f64(theta_i)
        assert th.shape == (self.ds.M, 3), th.shape
        o = self._opt(opt_kw)
        n0, ni = np.empty(2), np.empty((self.ds.M, 3))
        _lib.call("magma_m_step", self._ctx, self._ds, _hp(th0), _hp(th), int(bool(common_hp)), _hp(o),
                  _hp(n0), _hp(ni))
        return n0, ni

    def logL_monitoring(self, theta0, theta_i):
        th0 = _f64(theta0)
        th, stride = self._theta_i(theta_i)
        out = C.c_double(0.0)
        _lib.call("magma_logl_monitoring", self._ctx, self._ds, _hp(th0), _hp(th), stride, C.byref(out))
        return out.value

    def sum_fi(self, theta_i, want_grad=False):
        """Sum over ALL individuals of logL_GP_mod (and of gr_GP_mod): CF:203-234, 334-368 with one
        triple, CF:254-261 with [M, 3]. Returns (f, g[3])."""
        th, stride = self._theta_i(theta_i)
        out = np.zeros(4)
        _lib.call("magma_sum_fi", self._ctx, self._ds, _hp(th), stride, int(bool(want_grad)), _hp(out))
        return float(out[0]), out[1:].copy()

    def em_iteration(self, theta0, theta_i, m0, common_hp, opt_kw=None):
        th0, th = _f64(theta0), _f64(theta_i)
        assert th.shape == (self.ds.M, 3), th.shape
        mv, nm = self._m0(m0)
        o = self._opt(opt_kw)
        n0, ni, ll = np.empty(2), np.empty((self.ds.M, 3)), np.empty(2)
        try:
            _lib.call("magma_em_iteration", self._ctx, self._ds, _hp(th0), _hp(th), _hp(mv), nm,
                      int(bool(common_hp)), _hp(o), _hp(n0), _hp(ni), _hp(ll))
        except _lib.MagmaHipError as e:
            if "not positive definite" in str(e):
                raise np.linalg.LinAlgError(str(e)) from None
            raise
        return n0, ni, float(ll[0]), float(ll[1])

    def training(self, prior_mean, ini_theta0, ini_theta_i, common_hp=True, n_loop_max=25,
                 trace=None, opt_kw=None):
        mv, nm = self._m0(prior_mean)
        i0, i3 = _f64(ini_theta0), _f64(ini_theta_i)
        o = self._opt(opt_kw)
        n0, ni = np.empty(2), np.empty((self.ds.M, 3))
        cv, its = C.c_int(0), C.c_int(0)
        tr = np.full((n_loop_max, 3), np.nan)
        _lib.call("magma_training", self._ctx, self._ds, _hp(mv), nm, _hp(i0), _hp(i3),
                  int(bool(common_hp)), int(n_loop_max), _hp(o), _hp(n0), _hp(ni), C.byref(cv),
                  C.byref(its), _hp(tr))
        if trace is not None:
            for k in range(its.value):
                if not np.isnan(tr[k, 0]):
                    trace.append({"iter": k + 1, "logLL": tr[k, 0], "logL_new_hp": tr[k, 1],
                                  "eps": tr[k, 2]})
        return n0, ni, bool(cv.value)

    # ------------------------------------------------------------ results
    def set_mean_cov(self, mean, cov):
        mean, cov = _f64(mean), _f64(cov)
        _lib.call("magma_set_mean_cov", self._ctx, self._ds, _hp(mean), _hp(cov), cov.shape[0])

    def set_m0(self, m0):
        mv, nm = self._m0(m0)
        _lib.call("magma_set_m0", self._ctx, self._ds, _hp(mv), nm)

    def get_mean(self):
        out = np.empty(self.N)
        _lib.call("magma_get_mean", self._ctx, self._ds, _hp(out))
        return out

    def get_cov(self):
        out = np.empty((self.N, self.N))  # symmetric: column-major == row-major
        _lib.call("magma_get_cov", self._ctx, self._ds, _hp(out), self.N)
        return out

    def get_cov_diag(self):
        return np.diag(self.get_cov()).copy()

    def pred_gp_batched(self, off_n, tn, yn, off_t, ts, theta):
        """pred_gp (MG:135-181) for B new individuals against this engine's resident
        hyper-posterior (magma_pred_gp_resident). Same contract as HipBackend.pred_gp_batched."""
        off_n = np.ascontiguousarray(off_n, dtype=np.int64)
        off_t = np.ascontiguousarray(off_t, dtype=np.int64)
        B = off_n.size - 1
        tn, yn, ts = _f64(tn), _f64(yn), _f64(ts)
        th = _f64(theta)
        stride = 0 if th.ndim == 1 else 3
        mean, var = np.empty(max(ts.size, 1)), np.empty(max(ts.size, 1))
        info = np.zeros(max(B, 1), dtype=np.int32)
        try:
            _lib.call("magma_pred_gp_resident", self._ctx, self._ds, B, _hp(off_n), _hp(tn), _hp(yn),
                      _hp(off_t), _hp(ts), _hp(th), stride, _hp(mean), _hp(var), _hp(info))
        except _lib.MagmaHipError as e:
            if "subscript out of bounds" in str(e):
                raise KeyError("subscript out of bounds") from None
            raise
        return mean[:ts.size], var[:ts.size], info[:B]


def train_new_gp(t, y, mean, cov, ini3, opt_kw=None, device=None):
    """magma_train_new_gp on sorted (t, y) with the mean process / cov block at t (cov None = 0)."""
    import torch
    ctx = C.c_void_p(0)
    dev = torch.cuda.current_device() if device is None else int(device)
    _lib.call("magma_ctx_create", C.byref(ctx), dev)
    try:
        t, y, mean, ini3 = _f64(t), _f64(y), _f64(mean), _f64(ini3)
        o = NativeEngine._opt(opt_kw)
        out, st = np.empty(3), C.c_int(0)
        cv = _f64(cov) if cov is not None else None
        _lib.call("magma_train_new_gp", ctx, int(t.size), _hp(t), _hp(y), _hp(mean),
                  _hp(cv) if cv is not None else None, int(t.size), _hp(ini3), _hp(o), _hp(out),
                  C.byref(st))
    finally:
        _lib.load().magma_ctx_destroy(ctx)
    return out, st.value
