"""The reference's `tryCatch(solve(mat), error = function(e) MASS::ginv(mat))` (Computing_functions.R:
133,161; MAGMA.R:173) on the product side: per-individual covariances that the Cholesky-type
sweep flags (singular to rounding: sigma = 0 with a huge length scale, near-duplicate
timestamps) are answered by LU with partial pivoting or, when that is singular to machine
precision, by the pseudo-inverse -- inside the library, never through oracle/.

PARITY UNPINNED against R: no reference fixture reaches this path. The checker is the oracle's
restatement (numpy.linalg.inv + the rcond rule of R's solve(), numpy.linalg.pinv with MASS::ginv's
tolerance). A pseudo-inverse is only as well determined as its smallest kept singular value
(~ 1.5e-8 of the largest), so agreement is asked at 1e-5 relative, not at 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import magma_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import magma_amd
    return magma_amd


def _db(rng, sizes):
    grid = np.arange(0.0, 10.0001, 0.05).round(2)
    ids, ts, ys = [], [], []
    for k, n in enumerate(sizes):
        t = np.sort(rng.choice(grid, n, replace=False))
        ids += [k + 1] * n
        ts.append(t)
        ys.append(rng.normal(0, 2, n) + 0.5 * t)
    return {"ID": np.array(ids), "Timestamp": np.concatenate(ts), "Output": np.concatenate(ys)}


SINGULAR = {2: [5.0, 8.0, 0.0],      # sigma = 0, length scale e^4: K is numerically rank ~ 4
            4: [1.0, 6.0, 1e-9],     # tiny noise
            5: [0.0, 12.0, 0.0]}     # a matrix of (almost) ones


def test_kern_to_inv_of_singular_matrices_matches_solve_or_ginv(m):
    rng = np.random.default_rng(0)
    db = _db(rng, [30, 12, 40, 7, 25, 33, 2, 64])
    ids = orc.unique_in_order(db["ID"])
    theta = {i: np.array(SINGULAR.get(int(i), [1.0, 1.0, 0.4])) for i in ids}
    got = m.kern_to_inv(db, m.kernel, theta, 0)
    exp = orc.kern_to_inv(db, orc.kernel, theta, 0)
    n_ginv = 0
    for i in ids:
        K = orc.kern_to_cov(db["Timestamp"][db["ID"] == i], orc.kernel, theta[i][:2], theta[i][2])[1]
        g, e = got[i][1], exp[i][1]
        assert np.all(np.isfinite(g))
        if int(i) in SINGULAR:
            n_ginv += 1
            assert np.array_equal(g, g.T)  # the slow path stores a mirrored lower triangle
            assert np.linalg.cond(K) > 1e15  # really singular: the oracle took pinv
            assert np.abs(g - e).max() < 1e-5 * np.abs(e).max(), (i, np.abs(g - e).max() / np.abs(e).max())
            # Moore-Penrose conditions that survive the truncation: G K G = G, (K G)' = K G
            assert np.abs(g @ K @ g - g).max() < 1e-6 * np.abs(g).max()
            kg = K @ g
            assert np.abs(kg - kg.T).max() < 1e-6
        else:
            assert np.abs(g - e).max() < 1e-9 * np.abs(e).max()
    assert n_ginv == 3


@pytest.mark.parametrize("route", ["gather", "atomic"])
def test_e_step_with_singular_individuals_matches_oracle(m, route):
    """CF:383-391 with three individuals whose K_i is singular: the reference adds ginv(K_i) to
    the precision; so does the HIP path (both E-step routes), instead of dropping them."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(1)
    db = _db(rng, [30, 12, 40, 7, 25, 33, 2, 50])
    ds = Dataset(db)
    thi = np.array([SINGULAR.get(int(i), [1.0, 1.0, 0.4]) for i in ds.ids], dtype=float)
    th0 = np.array([0.9, 0.7])
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    pen = float(np.mean(thi[:, 2]))
    eng = Engine(ds)
    eng.be.scatter_route = route
    assert eng.e_step(th0, thi, pen, 0.4) == 0
    assert eng.be.n_flagged() == 3
    exp = orc.e_step(db, 0.4, orc.kernel, orc.kernel, hp)
    cov, mean = eng.be.get_cov(), eng.be.get_mean()
    # the singular individuals enter with huge precisions (1 / smallest kept eigenvalue), which
    # pin the posterior along a few directions: compare at the pseudo-inverse's own accuracy
    assert np.abs(cov - exp["cov"][1]).max() < 1e-5 * np.abs(exp["cov"][1]).max()
    assert np.abs(mean - exp["mean"]["Output"]).max() < 1e-5 * np.abs(exp["mean"]["Output"]).max()
    # and they DO count: leaving them out changes the answer by much more than that
    keep = ~np.isin(db["ID"], list(SINGULAR))
    db2 = {k: v[keep] for k, v in db.items()}
    hp2 = {"theta_0": th0, "theta_i": {i: v for i, v in hp["theta_i"].items() if int(i) not in SINGULAR}}
    grid = np.unique(db["Timestamp"])
    without = orc._hyperposterior(db2, grid, 0.4, orc.kernel, orc.kernel, hp2, pen)
    assert np.abs(without["cov"][1] - exp["cov"][1]).max() > 1e-2 * np.abs(exp["cov"][1]).max()


def test_native_driver_counts_and_includes_flagged_individuals(m):
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(2)
    db = _db(rng, [30, 12, 40, 7, 25])
    ds = m.Dataset(db)
    thi = np.array([SINGULAR.get(int(i), [1.0, 1.0, 0.4]) for i in ds.ids], dtype=float)
    th0 = np.array([0.9, 0.7])
    nat = NativeEngine(ds)
    assert nat.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.0) == 0
    assert nat.stats["flagged"] == 3
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    exp = orc.e_step(db, 0.0, orc.kernel, orc.kernel, hp)
    assert np.abs(nat.get_cov() - exp["cov"][1]).max() < 1e-5 * np.abs(exp["cov"][1]).max()
    nat.close()
