"""BASELINE.json's configurations at full size, through size-independent properties (the
oracle would take minutes to hours there): inverse round trips, additivity of the sharded
scatter, consistency between launch variants, finite differences; plus config 2 (common grid,
common HP), which is small enough for a direct oracle comparison."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import magma_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    return torch


def _engine(M, n, N, seed, **kw):
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    from magma_amd.synth import simu_arrays
    a = simu_arrays(M, n, N, seed=seed, **kw)
    ds = Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    return a, ds, Engine(ds)


def test_c3_estep_round_trip_and_shard_additivity(gpu):
    torch = gpu
    from magma_amd.engine import HipBackend
    M, n, N = 10000, 50, 2000
    a, ds, eng = _engine(M, n, N, 1003)
    rng = np.random.default_rng(0)
    thi = np.column_stack([rng.uniform(0, 3, M), rng.uniform(0, 2, M), rng.uniform(0.2, 2, M)])
    th0, pen, m0 = np.array([1.5, 0.8]), 0.9, 0.3
    be = eng.be
    P, w = be.estep_scatter(thi)
    Np = be.Np
    Pl = np.tril(P.view(Np, Np).t().cpu().numpy()[:N, :N])  # scatter fills the lower triangle
    P_full = Pl + np.tril(Pl, -1).T
    w_full = w[:N].cpu().numpy()
    # additivity: two half shards on the same GPU sum to the full scatter (all-reduce step)
    acc_P, acc_w = np.zeros_like(Pl), np.zeros(N)
    for lo, hi in ((0, M // 2), (M // 2, M)):
        b2 = HipBackend(ds, lo, hi)
        P2, w2 = b2.estep_scatter(thi[lo:hi])
        acc_P += np.tril(P2.view(Np, Np).t().cpu().numpy()[:N, :N])
        acc_w += w2[:N].cpu().numpy()
        del b2
    assert np.abs(acc_P - Pl).max() < 1e-11 * np.abs(Pl).max()
    assert np.abs(acc_w - w_full).max() < 1e-11 * np.abs(w_full).max()
    # finish: cov must invert (K0^-1 + scatter), mean = cov (K0^-1 m0 + w)
    be.set_m0(m0)
    assert be.estep_finish(th0, pen)[0] == 0
    cov, mean = be.get_cov(), be.get_mean()
    K0inv = np.linalg.inv(orc.kern_to_cov(ds.grid, orc.kernel, th0, pen)[1])
    prec = P_full + K0inv
    resid = prec @ cov - np.eye(N)
    assert np.abs(resid).max() < 1e-9
    assert np.abs(cov - cov.T).max() == 0.0
    rhs = K0inv @ np.full(N, m0) + w_full
    assert np.abs(prec @ mean - rhs).max() < 1e-8 * np.abs(rhs).max()
    # linearity of update_mean in y: scaling the outputs scales w
    be.y.mul_(2.0)
    _, w2 = be.estep_scatter(thi)
    assert np.abs(w2[:N].cpu().numpy() - 2.0 * w_full).max() < 1e-11 * np.abs(w_full).max()
    be.y.mul_(0.5)


def test_c3_objective_launch_variants_and_finite_differences(gpu):
    M, n, N = 10000, 50, 2000
    a, ds, eng = _engine(M, n, N, 1003)
    thi = np.tile([1.0, 1.0, 0.2], (M, 1))
    eng.e_step(np.array([1.0, 1.0]), thi, 0.2, 0.0)
    rng = np.random.default_rng(1)
    th = np.column_stack([rng.uniform(0, 3, M), rng.uniform(0, 2, M), rng.uniform(0.3, 2, M)])
    f_all, g_all = eng.be.indiv_eval(th, True)
    f_only, _ = eng.be.indiv_eval(th, False)
    assert np.all(np.isfinite(f_all)) and np.all(np.isfinite(g_all))
    assert np.abs(f_all - f_only).max() < 1e-11 * np.abs(f_all).max()
    # compact active list == full launch on those individuals
    act = np.zeros(M, bool)
    act[rng.choice(M, 137, replace=False)] = True
    f_sel, g_sel = eng.be.indiv_eval(th, True, active=act)
    assert np.array_equal(f_sel[act], f_all[act]) and np.array_equal(g_sel[act], g_all[act])
    assert np.all(f_sel[~act] == 0)
    # central finite differences on all individuals at once (3 extra pairs of launches)
    h = 1e-6
    for k in range(3):
        e = np.zeros(3)
        e[k] = h
        fp, _ = eng.be.indiv_eval(th + e, False)
        fm, _ = eng.be.indiv_eval(th - e, False)
        fd = (fp - fm) / (2 * h)
        err = np.abs(fd - g_all[:, k]) / np.maximum(1.0, np.abs(g_all[:, k]))
        assert np.quantile(err, 0.99) < 1e-5 and err.max() < 1e-3, (k, err.max())
    # theta_0 objective: gradient vs finite differences, f-only == f of the f+g call
    t0 = np.array([1.3, 0.7])
    f, g, info = eng.be.theta0_eval(t0, 0.5, True)
    f2, _, _ = eng.be.theta0_eval(t0, 0.5, False)
    assert info == 0 and abs(f - f2) < 1e-12 * abs(f)
    for k in range(2):
        e = np.zeros(2)
        e[k] = 1e-5
        fd = (eng.be.theta0_eval(t0 + e, 0.5, False)[0] - eng.be.theta0_eval(t0 - e, 0.5, False)[0]) / 2e-5
        assert abs(fd - g[k]) < 1e-5 * max(1.0, abs(g[k])), (k, fd, g[k])


def test_c2_common_grid_common_hp_matches_oracle(gpu):
    """Config 2: M = 1,000 x 50 shared timestamps, common hyper-parameters."""
    import magma_amd as m
    from magma_amd.synth import simu_db
    db, a = simu_db(1000, 50, 50, common_times=True, common_hp=True, seed=1002)
    ids = orc.unique_in_order(db["ID"])
    th = np.array([1.0, 1.0, 0.2])
    hp = {"theta_0": np.array([1.0, 1.0]), "theta_i": {i: th for i in ids}}
    got = m.e_step(db, 0.0, m.kernel_mu, m.kernel, hp)
    exp = orc.e_step(db, 0.0, orc.kernel, orc.kernel, hp)
    assert np.abs(got["cov"][1] - exp["cov"][1]).max() < 1e-9 * np.abs(exp["cov"][1]).max()
    assert np.abs(got["mean"]["Output"] - exp["mean"]["Output"]).max() < 1e-9 * np.abs(exp["mean"]["Output"]).max()
    th2 = np.array([1.4, 0.6, 0.5])
    f = m.logL_GP_mod_common_hp(th2, db, exp["mean"], m.kernel, exp["cov"])
    g = m.gr_GP_mod_common_hp(th2, db, exp["mean"], m.kernel, exp["cov"])
    f_o = orc.logL_GP_mod_common_hp(th2, db, exp["mean"], orc.kernel, exp["cov"])
    g_o = orc.gr_GP_mod_common_hp(th2, db, exp["mean"], orc.kernel, exp["cov"])
    assert abs(f - f_o) < 1e-10 * abs(f_o)
    assert np.abs(g - g_o).max() < 1e-8 * np.abs(g_o).max()


@pytest.mark.parametrize("N", [4000, 8000], ids=["tile-graph-63-block-rows", "launch-per-step"])
def test_c4_large_grid_inverse_round_trip(gpu, N):
    """Config 4: N = 8,000 union inputs -- dense MFMA Cholesky / inverse at full size (the
    launch-per-step path the library picks above 64 block rows) and N = 4,000 (the largest size
    of the one-launch tile graph), checked by A A^-1 = I with an independent FP64 product
    (torch), plus the log-det against torch."""
    torch = gpu
    import ctypes as C
    from magma_amd import _lib
    lib = _lib.load()
    Np = lib.magma_pad(N)
    g = torch.linspace(0, 10, N, dtype=torch.float64, device="cuda")
    gp = torch.zeros(Np, dtype=torch.float64, device="cuda")
    gp[:N] = g
    A = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    _lib.call("magma_kern_to_cov_grid", N, p(gp), 1.0, 0.5, 0.7, p(A), st)
    K = A.view(Np, Np)[:N, :N].clone()
    work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
    ld = torch.zeros(1, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.call("magma_spd_inverse", N, p(A), p(ld), p(info), p(work), st)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    Kinv = A.view(Np, Np)[:N, :N]
    resid = (K @ Kinv - torch.eye(N, dtype=torch.float64, device="cuda")).abs().max().item()
    assert resid < 1e-8, resid
    assert abs(ld.item() - torch.linalg.slogdet(K)[1].item()) < 1e-8 * abs(ld.item())


def test_c5_scatter_at_100k_individuals(gpu):
    """Config 5's per-GPU shape at 8 GPUs is 12,500 x 40; here the whole M = 100,000 on one GPU:
    the scatter's trace identity sum_i tr(K_i^-1) == tr(P) and flag-free completion."""
    M, n, N = 100000, 40, 2000
    a, ds, eng = _engine(M, n, N, 1005)
    thi = np.tile([1.0, 1.0, 0.5], (M, 1))
    P, w = eng.be.estep_scatter(thi)
    assert eng.be.n_flagged() == 0
    trP = float(P.view(eng.be.Np, eng.be.Np).diagonal().sum().item())
    # with identical HPs, tr(K_i^-1) depends on t_i only: check a sample against the oracle
    idx = np.random.default_rng(0).choice(M, 200, replace=False)
    sample = [np.trace(orc.kern_to_inv(a["t"][i * n:(i + 1) * n], orc.kernel, [1.0, 1.0], 0.5)[1]) for i in idx]
    assert abs(trP / M - np.mean(sample)) < 5 * np.std(sample) / np.sqrt(len(sample))
    assert np.isfinite(w.sum().item())


def test_c3_mstep_routes_agree_and_decrease_the_objective(gpu):
    """Config 3 at full size: the device-resident M-step and the library's lock-step loop end
    every one of the 10^4 problems in the same state, and no problem ends above its start."""
    M, n, N = 10000, 50, 2000
    a, ds, eng = _engine(M, n, N, 1003)
    th0 = np.array([1.0, 1.0])
    thi = np.tile([1.0, 1.0, 0.2], (M, 1))
    assert eng.e_step(th0, thi, 0.2, 0.0) == 0
    from magma_amd import _lib
    s = eng.be.indiv_mstep(thi)
    # Problems with a trial point whose covariance the sweep flags: the lock-step route evaluates
    # those through solve() / ginv() (CF:133), the resident kernel bisects there by default. With
    # "resident_slow_path" it marks exactly those problems (status 6) instead.
    try:
        _lib.call("magma_set_option", b"resident_slow_path", 1)
        r6 = eng.be.indiv_mstep_resident(thi)
    finally:
        _lib.call("magma_set_option", b"resident_slow_path", 0)
    r = eng.be.indiv_mstep_resident(thi)
    hit = r6["status"] == 6
    same = ~hit
    print(f"C3 first M-step: {int(hit.sum())} of {M} problems meet a flagged trial point")
    assert hit.sum() < 0.01 * M
    for key in ("status", "nfev", "iters"):
        assert np.array_equal(r[key][same], s[key][same]) and np.array_equal(r6[key][same], s[key][same])
    assert np.abs(r["x"][same] - s["x"][same]).max() <= 1e-10 * max(1.0, np.abs(s["x"]).max())
    assert np.abs(r["f"][same] - s["f"][same]).max() <= 1e-12 * np.abs(s["f"]).max()
    # the others end at the same optimum by either rule, to the optimiser's tolerance
    assert np.abs(r["f"][hit] - s["f"][hit]).max() <= 1e-6 * np.abs(s["f"][hit]).max()
    ok = np.isin(r["status"], (1, 2, 3, 4))
    assert ok.all() and np.mean(r["status"] == 1) > 0.95  # converged by relative reduction
    assert np.all(r["f"] <= r["f_x0"] + 1e-9 * np.abs(r["f_x0"]))
    # the final points are stationary to the optimiser's tolerance: re-evaluate the gradient
    f, g = eng.be.indiv_eval(r["x"], True)
    assert np.abs(f - r["f"]).max() <= 1e-10 * np.abs(f).max()
    conv = r["status"] == 1
    assert np.median(np.abs(g[conv]).max(axis=1)) < 1e-2
