"""Lock-step batched L-BFGS-B (the host-side optimiser in libmagma_hip.so, lbfgsb.cpp).

Stands in for `optimr::opm(par, fn, gr, method = "L-BFGS-B", control = list(kkt = FALSE))`
(Computing_functions.R:412-413, 417-418, 426-429; MAGMA.R:93-94) with stats::optim's defaults.
"""
import ctypes as C
import numpy as np
from . import _lib


def minimize_batch(x0, evaluate, m=5, factr=1e7, pgtol=0.0, maxit=100, max_rounds=100000):
    """Minimise B independent d-dimensional problems together.

    evaluate(x[B, d], active[B] bool) -> (f[B], g[B, d]); entries of inactive problems are
    ignored. Returns dict(x, f, iters, nfev, status, rounds); status 1 = converged on relative
    reduction, 2 = gradient tolerance, 3 = maxit, 4 = line search failed, 5 = non-finite f
    (x is NaN for those, mirroring R's error path).
    """
    lib = _lib.load()
    x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
    B, d = x0.shape
    h = lib.magma_lbfgsb_create(B, d, x0.ctypes.data, m, float(factr), float(pgtol), int(maxit))
    if not h:
        raise _lib.MagmaHipError("magma_lbfgsb_create failed")
    try:
        x = np.empty((B, d))
        active = np.empty(B, dtype=np.uint8)
        nact = C.c_int(0)
        rounds = 0
        f_first = None
        while True:
            _lib.check(lib.magma_lbfgsb_ask(h, x.ctypes.data, active.ctypes.data, C.byref(nact)))
            if nact.value == 0 or rounds >= max_rounds:
                break
            f, g = evaluate(x, active.astype(bool))
            f = np.ascontiguousarray(f, dtype=np.float64)
            g = np.ascontiguousarray(g, dtype=np.float64)
            if f_first is None:
                f_first = f.copy()  # objective at x0 (every problem is active in round 1)
            _lib.check(lib.magma_lbfgsb_tell(h, f.ctypes.data, g.ctypes.data))
            rounds += 1
        xr = np.empty((B, d))
        fr = np.empty(B)
        it = np.empty(B, dtype=np.int32)
        nf = np.empty(B, dtype=np.int32)
        stt = np.empty(B, dtype=np.int32)
        _lib.check(lib.magma_lbfgsb_result(h, xr.ctypes.data, fr.ctypes.data, it.ctypes.data,
                                           nf.ctypes.data, stt.ctypes.data))
    finally:
        lib.magma_lbfgsb_destroy(h)
    return {"x": xr, "f": fr, "iters": it, "nfev": nf, "status": stt, "rounds": rounds,
            "f_x0": f_first}
