"""The reference's `tryCatch(solve(.), error = ginv(.))` inside the M-step's objectives and the
prediction (Computing_functions.R:133 via logL_GP_mod / gr_GP_mod, :177, :285 for logL_GP / gr_GP;
MAGMA.R:173 for pred_gp): slots whose covariance the sweep flags are evaluated from solve() /
ginv() of that covariance by indiv_eval_slowpath_kernel / pred_gp_*_slowpath_kernel, inside the
library -- never through oracle/.

PARITY UNPINNED against R: no reference fixture reaches this path. The checker is the oracle's
restatement of the same formulas (numpy inv / pinv, slogdet of the inverse). A pseudo-inverse is
determined to ~1e-5 relative (smallest kept singular value 1.5e-8 of the largest) and the
log-determinant of a rank-deficient pseudo-inverse is rounding noise in both implementations
(dmvnorm, CF:27, takes it from the LU of `inv`), so agreement is asked at 1e-4 relative of the
value's own scale, not at 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import magma_oracle as orc  # noqa: E402  (checker only)
from test_gpu_slowpath import _db, SINGULAR  # noqa: E402


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import magma_amd
    return magma_amd


def _setup(m, seed, sizes, th0=(0.9, 0.7), pen=0.4):
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(seed)
    db = _db(rng, sizes)
    ds = Dataset(db)
    good = np.tile([1.0, 1.0, 0.4], (ds.M, 1))
    th0 = np.array(th0)
    eng = Engine(ds)
    assert eng.e_step(th0, good, pen, 0.3) == 0  # a well-conditioned hyper-posterior to evaluate against
    hp = {"theta_0": th0, "theta_i": {i: good[k] for k, i in enumerate(ds.ids)}}
    exp = orc._hyperposterior(db, np.unique(db["Timestamp"]), 0.3, orc.kernel, orc.kernel, hp, pen)
    return db, ds, eng, exp


def _oracle_args(db, exp, i):
    sel = db["ID"] == i
    db_i = {"Timestamp": db["Timestamp"][sel], "Output": db["Output"][sel]}
    labels, covm = exp["cov"]
    return db_i, orc._mean_at(exp["mean"], db_i["Timestamp"]), orc._block(labels, covm, db_i["Timestamp"])


@pytest.mark.parametrize("mode,tiny", [(0, False), (1, False), (1, True)],
                         ids=["logL_GP_mod", "logL_GP", "logL_GP-tiny-cov"])
def test_flagged_evaluations_take_solve_or_ginv(m, mode, tiny):
    """tiny: a hyper-posterior covariance of ~1e-18 (a prior variance of e^-40), so that K + C of
    logL_GP (CF:176) is as singular as K alone and the slow path adds C itself."""
    db, ds, eng, exp = _setup(m, 3, [30, 12, 40, 7, 25, 33, 2, 64],
                              *(((-40.0, 0.7), 1e-9) if tiny else ()))
    thi = np.array([SINGULAR.get(int(i), [1.0, 1.0, 0.4]) for i in ds.ids], dtype=float)
    f, g = eng.be.indiv_eval(thi, True, mode=mode)
    info = eng.be.info[:ds.M].cpu().numpy()
    fo_only, _ = eng.be.indiv_eval(thi, False, mode=mode)
    worst = {"f": 0.0, "g": 0.0, "logdet": 0.0}
    nflag = 0
    theta_d = {i: thi[k] for k, i in enumerate(ds.ids)}
    inv_gpu = m.kern_to_inv(db, m.kernel, theta_d, 0)  # the library's own solve() / ginv() answers
    for k, i in enumerate(ds.ids):
        db_i, mean_i, cov_i = _oracle_args(db, exp, i)
        fn, gr = (orc.logL_GP_mod, orc.gr_GP_mod) if mode == 0 else (orc.logL_GP, orc.gr_GP)
        fo, go = fn(thi[k], db_i, mean_i, orc.kernel, cov_i), gr(thi[k], db_i, mean_i, orc.kernel, cov_i)
        assert np.isfinite(f[k]) and np.all(np.isfinite(g[k])), (i, f[k], g[k])
        assert abs(fo_only[k] - f[k]) <= 1e-13 * abs(f[k])  # (the f-only kernel sums in another order)
        if mode == 0 and int(i) in SINGULAR:
            assert info[k] != 0
        eg = np.abs(g[k] - go).max() / max(1.0, np.abs(go).max())
        if info[k] == 0:
            assert abs(f[k] - fo) < 1e-9 * max(1.0, abs(fo)) and eg < 1e-9, (i, f[k], fo, g[k], go)
            continue
        nflag += 1
        # The gradient has no determinant in it: 1e-4 of its scale (measured 1e-10).
        assert eg < 1e-4, (i, g[k], go)
        # f = log-determinant term + rest. The log-determinant of a rank-deficient pseudo-inverse
        # is rounding noise in BOTH implementations (its n - r null pivots are O(eps) numbers,
        # dmvnorm takes their logs, CF:27): the rest is compared at 1e-4, the term is asked to be
        # finite and within a factor 1e4 per null pivot of the oracle's.
        K = orc.kern_to_cov(db_i["Timestamp"], orc.kernel, thi[k][:2], thi[k][2])[1]
        if mode == 1:
            K = K + cov_i
        inv_o = orc._solve_or_ginv(K)
        z = db_i["Output"] - mean_i
        n = z.size
        sv = np.linalg.svd(K, compute_uv=False)
        null = int(np.sum(sv <= np.sqrt(np.finfo(float).eps) * sv[0]))
        rest_o = 0.5 * (n * orc.LOG_2PI + z @ inv_o @ z) + (0.5 * np.sum(inv_o * cov_i) if mode == 0 else 0.0)
        term_o = fo - rest_o
        if mode == 0:
            R = inv_gpu[i][1]
            rest_g = 0.5 * (n * orc.LOG_2PI + z @ R @ z) + 0.5 * np.sum(R * cov_i)
            assert abs(rest_g - rest_o) < 1e-4 * max(1.0, abs(rest_o)), (i, rest_g, rest_o)
        term_g = f[k] - rest_o
        assert np.isfinite(term_g)
        assert abs(term_g - term_o) <= 0.5 * np.log(1e4) * max(null, 1) + 1e-4 * abs(rest_o), (i, null, term_g, term_o)
        ef = abs(f[k] - fo) / max(1.0, abs(fo))
        worst["f"], worst["g"] = max(worst["f"], ef), max(worst["g"], eg)
        worst["logdet"] = max(worst["logdet"], abs(term_g - term_o))
    print(f"mode {mode}: {nflag} flagged slots, worst relative difference f {worst['f']:.1e}, g {worst['g']:.1e}; "
          f"log-determinant terms differ by up to {worst['logdet']:.2f}")
    # (mode 1 adds the hyper-posterior block, which regularises K unless it is tiny)
    assert nflag == (3 if (mode == 0 or tiny) else 0)


def test_m_step_routes_flagged_problems_through_the_lock_step_slow_path(m):
    """magma_m_step: an individual whose start is a singular covariance stops the resident kernel
    with status 6 and is solved again by the lock-step route (every evaluation through solve() /
    ginv() where flagged). The objective the library reports at its answer is the oracle's."""
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(4)
    db = _db(rng, [30, 12, 40, 7, 25])
    ds = m.Dataset(db)
    th0 = np.array([0.9, 0.7])
    good = np.tile([1.0, 1.0, 0.4], (ds.M, 1))
    nat = NativeEngine(ds)
    assert nat.e_step(th0, good, 0.4, 0.3) == 0
    hp = {"theta_0": th0, "theta_i": {i: good[k] for k, i in enumerate(ds.ids)}}
    exp = orc._hyperposterior(db, np.unique(db["Timestamp"]), 0.3, orc.kernel, orc.kernel, hp, 0.4)
    start = np.array([SINGULAR.get(int(i), [1.0, 1.0, 0.4]) for i in ds.ids], dtype=float)
    nat.reset_stats()
    n0, ni = nat.m_step(th0, start, False)
    st = nat.stats
    assert st["slow_resolved"] >= 1, st
    assert np.all(np.isfinite(ni)) and np.all(np.isfinite(n0))
    # unflagged problems are untouched by the re-solve: same answer as without singular starts
    n0b, nib = nat.m_step(th0, good, False)
    for k, i in enumerate(ds.ids):
        if int(i) not in SINGULAR:
            assert np.array_equal(ni[k], nib[k])
    # the re-solved ones: the optimiser went downhill from the start, judged by the oracle
    for k, i in enumerate(ds.ids):
        if int(i) in SINGULAR:
            db_i, mean_i, cov_i = _oracle_args(db, exp, i)
            f_start = orc.logL_GP_mod(start[k], db_i, mean_i, orc.kernel, cov_i)
            f_end = orc.logL_GP_mod(ni[k], db_i, mean_i, orc.kernel, cov_i)
            assert f_end <= f_start + 1e-6 * abs(f_start), (i, f_start, f_end)
    nat.close()


def test_pred_gp_with_a_singular_covariance_takes_solve_or_ginv(m):
    """MG:173: inv_mat = tryCatch(solve(cov_tn_tn), ginv). sigma = 0 with a huge length scale and
    no hyper-posterior covariance makes cov_tn_tn singular."""
    rng = np.random.default_rng(5)
    t = np.sort(rng.choice(np.arange(0.0, 10.0001, 0.05).round(2), 30, replace=False))
    db = {"Timestamp": t, "Output": rng.normal(0, 1, 30) + 0.3 * t}
    ts = np.linspace(0.0, 10.0, 41)
    got = m.pred_gp(db, ts, 0.5, None, m.kernel, [5.0, 8.0], 0.0)
    exp = orc.pred_gp(db, ts, 0.5, None, orc.kernel, [5.0, 8.0], 0.0)
    assert np.linalg.cond(orc.kern_to_cov(t, orc.kernel, [5.0, 8.0], 0.0)[1]) > 1e15
    sm = np.abs(exp["Mean"]).max()
    sv = max(np.abs(exp["Var"]).max(), np.exp(5.0))
    assert np.abs(got["Mean"] - exp["Mean"]).max() < 1e-4 * sm, np.abs(got["Mean"] - exp["Mean"]).max() / sm
    assert np.abs(got["Var"] - exp["Var"]).max() < 1e-4 * sv, np.abs(got["Var"] - exp["Var"]).max() / sv


def test_pred_gp_batched_with_singular_covariances_takes_solve_or_ginv(m):
    """The device-resident prediction of three new individuals, two of them with a
    covariance that is singular to rounding (hyper-posterior covariance ~1e-18, sigma = 0): their
    answers come from solve() / ginv() (MG:173), the third from the sweep as before."""
    rng = np.random.default_rng(6)
    db = _db(rng, [20, 15, 30])
    ids = orc.unique_in_order(db["ID"])
    hp = {"theta_0": np.array([-40.0, 0.7]), "theta_i": {i: np.array([1.0, 1.0, 0.4]) for i in ids}}
    grid = np.arange(0.0, 10.0001, 0.05).round(2)
    new_dbs, tss = [], []
    for n in (25, 12, 18):
        t = np.sort(rng.choice(grid, n, replace=False))
        new_dbs.append({"Timestamp": t, "Output": rng.normal(0, 1, n) + 0.3 * t})
        tss.append(np.sort(rng.choice(grid, 9, replace=False)))
    thetas = np.array([[5.0, 8.0], [1.0, 0.5], [0.0, 12.0]])
    sigmas = np.array([0.0, 0.3, 0.0])
    eng = m.posterior_mu_resident(db, new_dbs, np.concatenate(tss), 0.2, m.kernel_mu, m.kernel, hp, pen_diag=1e-9)
    got = m.pred_gp_batched(eng, new_dbs, tss, thetas, sigmas)
    all_new = {"Timestamp": np.concatenate([d["Timestamp"] for d in new_dbs])}
    mu = orc.posterior_mu(db, all_new, np.concatenate(tss), 0.2, orc.kernel, orc.kernel, hp, pen_diag=1e-9)
    for b, nd in enumerate(new_dbs):
        exp = orc.pred_gp(nd, tss[b], mu["mean"], mu["cov"], orc.kernel, thetas[b], sigmas[b])
        tol = 1e-8 if b == 1 else 1e-4
        sm = max(1.0, np.abs(exp["Mean"]).max())
        sv = max(1.0, np.abs(exp["Var"]).max(), np.exp(thetas[b][0]))
        assert np.abs(got[b]["Mean"] - exp["Mean"]).max() < tol * sm, (b, np.abs(got[b]["Mean"] - exp["Mean"]).max() / sm)
        assert np.abs(got[b]["Var"] - exp["Var"]).max() < tol * sv, (b, np.abs(got[b]["Var"] - exp["Var"]).max() / sv)
