"""CPU-only tests: the C ABI library loads and exports what include/magma_hip.h declares, the
host-side L-BFGS-B restatement against scipy's, the Dataset layout, and the oracle's own
consistency checks (finite differences, E-step == posterior_mu identity, n = 1)."""
import os
import re
import ctypes
import numpy as np
import pytest

from oracle import magma_oracle as orc
from kat_common import dataset, hp_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from magma_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "magma_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(magma_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    assert names == set(_lib.SIGNATURES), names ^ set(_lib.SIGNATURES)
    assert _lib.load().magma_pad(2000) == 2048 and _lib.load().magma_pad(64) == 64


def test_lbfgsb_matches_scipy_on_rosenbrock_batch():
    from scipy.optimize import minimize, rosen, rosen_der
    from magma_amd.lbfgsb import minimize_batch
    rng = np.random.default_rng(0)
    x0 = rng.uniform(-2, 2, size=(8, 3))

    def ev(x, act):
        return np.array([rosen(v) for v in x]), np.array([rosen_der(v) for v in x])
    res = minimize_batch(x0, ev, maxit=100)
    for b in range(8):
        r = minimize(rosen, x0[b], jac=rosen_der, method="L-BFGS-B",
                     options={"maxcor": 5, "ftol": 1e7 * np.finfo(float).eps, "gtol": 0.0,
                              "maxiter": 100, "maxls": 20})
        assert res["nfev"][b] == r.nfev and res["iters"][b] == r.nit, (b, res["nfev"][b], r.nfev)
        assert np.abs(res["x"][b] - r.x).max() < 1e-6
    assert np.all(res["status"] == 1)


def test_lbfgsb_recovers_from_non_finite_trial_points():
    from magma_amd.lbfgsb import minimize_batch

    calls = {"nan": 0}

    def ev(x, act):  # flat quadratic with its minimum next to a wall of NaN at x >= 3
        bad = x[:, 0] >= 3.0
        calls["nan"] += int(bad.sum())
        f = np.where(bad, np.nan, 1e-4 * ((x - 2.9) ** 2).sum(axis=1))
        return f, np.where(bad[:, None], np.nan, 2e-4 * (x - 2.9))
    res = minimize_batch(np.array([[2.5, 2.5], [1.0, 2.0]]), ev)
    assert calls["nan"] > 0  # the first (unit-length) trial step lands behind the wall
    assert np.all(np.isin(res["status"], (1, 2))) and np.abs(res["x"] - 2.9).max() < 1e-3
    res = minimize_batch(np.array([[5.0, 0.0]]), ev)  # cannot even start: R errors, we flag
    assert res["status"][0] == 5 and np.all(np.isnan(res["x"]))


# ----------------------------------------------------------------- oracle self-consistency
def _fd(fn, x, h=1e-6):
    g = np.zeros_like(x)
    for k in range(x.size):
        e = np.zeros_like(x)
        e[k] = h
        g[k] = (fn(x + e) - fn(x - e)) / (2 * h)
    return g


@pytest.mark.parametrize("n", [1, 2, 12])
def test_oracle_gradients_match_finite_differences(n):
    rng = np.random.default_rng(n)
    t = np.sort(rng.choice(np.arange(0, 10, 0.05), n, replace=False))
    db = {"Timestamp": t, "Output": rng.normal(size=n)}
    B = rng.normal(size=(n, n))
    C = B @ B.T / n
    mean = rng.normal(size=n)
    hp = np.array([0.7, 0.3, 0.6])
    for fn, gr in ((orc.logL_GP_mod, orc.gr_GP_mod), (orc.logL_GP, orc.gr_GP)):
        g = gr(hp, db, mean, orc.kernel, C)
        g_fd = _fd(lambda p: fn(p, db, mean, orc.kernel, C), hp)
        assert np.abs(g - g_fd).max() < 1e-5 * max(1.0, np.abs(g).max())
    g = orc.gr_GP_mod(hp[:2], db, mean, orc.kernel, C, pen_diag=0.5)
    g_fd = _fd(lambda p: orc.logL_GP_mod(p, db, mean, orc.kernel, C, pen_diag=0.5), hp[:2])
    assert np.abs(g - g_fd).max() < 1e-5 * max(1.0, np.abs(g).max())


def test_oracle_estep_equals_posterior_mu_on_training_grid():
    _, test, train = dataset("data_FF", 2)
    hp = hp_of("hp_FF", 2)
    e = orc.e_step(train, 0.0, orc.kernel, orc.kernel, hp)
    empty = {"ID": np.zeros(0), "Timestamp": np.zeros(0), "Output": np.zeros(0)}
    p = orc.posterior_mu(train, empty, np.zeros(0), 0.0, orc.kernel, orc.kernel, hp, "mean_sigma")
    assert np.array_equal(e["mean"]["Timestamp"], p["mean"]["Timestamp"])
    assert np.abs(e["cov"][1] - p["cov"][1]).max() == 0.0


@pytest.mark.parametrize("d", [1, 4])
def test_oracle_training_reproduces_stored_hyperparameters(d):
    """Full EM (`training`, MAGMA.R:15-82) with scipy's L-BFGS-B against the HPs R stored in
    train_FT.rds: pins e_step / m_step / logL_* / gr_* / the stopping rule of the oracle."""
    _, _, train = dataset("data_FT", d)
    res = orc.training(train, 0.0, {"theta_0": [1.0, 1.0], "theta_i": [1.0, 1.0, 0.2]},
                       orc.kernel, orc.kernel, common_hp=True)
    ref = hp_of("hp_FT", d)
    assert np.abs(res["hp"]["theta_0"] - ref["theta_0"]).max() < 1e-8
    assert max(np.abs(res["hp"]["theta_i"][i] - v).max() for i, v in ref["theta_i"].items()) < 1e-8


def test_lbfgsb_compact_ask_tell_equals_full_size_ask_tell():
    """The O(running problems) interface used by magma_indiv_mstep walks the same states as the
    full-size one (host only: no GPU call)."""
    import ctypes as C
    from magma_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    B, d = 40, 3
    x0 = rng.normal(0.0, 1.0, (B, d))
    scale = rng.uniform(0.5, 30.0, (B, d))
    ctr = rng.normal(0.0, 2.0, (B, d))

    def fg(x, rows):  # separable quartic + quadratic, different conditioning per problem
        z = x - ctr[rows]
        f = np.sum(scale[rows] * z ** 2 + 0.1 * z ** 4, axis=1)
        return f, 2 * scale[rows] * z + 0.4 * z ** 3

    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    out = []
    for compact in (False, True):
        s = lib.magma_lbfgsb_create(B, d, vp(np.ascontiguousarray(x0)), 5, 1e7, 0.0, 100)
        rounds = 0
        while True:
            na = C.c_int(0)
            if compact:
                sel = np.zeros(B, dtype=np.int32)
                x = np.zeros((B, d))
                lib.magma_lbfgsb_ask_compact(s, vp(sel), vp(x), C.byref(na))
                if na.value == 0:
                    break
                rows = sel[:na.value]
                f, g = fg(x[:na.value], rows)
                lib.magma_lbfgsb_tell_compact(s, na.value, vp(sel), vp(np.ascontiguousarray(f)),
                                              vp(np.ascontiguousarray(g)))
            else:
                act = np.zeros(B, dtype=np.uint8)
                x = np.zeros((B, d))
                lib.magma_lbfgsb_ask(s, vp(x), vp(act), C.byref(na))
                if na.value == 0:
                    break
                f, g = fg(x, np.arange(B))
                lib.magma_lbfgsb_tell(s, vp(np.ascontiguousarray(f)), vp(np.ascontiguousarray(g)))
            rounds += 1
            assert rounds < 5000
        xr, fr = np.zeros((B, d)), np.zeros(B)
        it, nf, stt = (np.zeros(B, dtype=np.int32) for _ in range(3))
        lib.magma_lbfgsb_result(s, vp(xr), vp(fr), vp(it), vp(nf), vp(stt))
        lib.magma_lbfgsb_destroy(s)
        out.append((xr, fr, it, nf, stt, rounds))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.abs(out[0][0] - ctr).max() < 1e-4


# ---------------------------------------------------------------- input layer
def _loop_grouping(db):
    """The reference's way: one `filter(ID == i)` scan per individual, timestamps sorted."""
    ids_col, t_col, y_col = np.asarray(db["ID"]), np.asarray(db["Timestamp"]), np.asarray(db["Output"])
    ids = list(orc.unique_in_order(ids_col))
    t, y, off = [], [], [0]
    for i in ids:
        sel = np.nonzero(ids_col == i)[0]
        o = np.argsort(t_col[sel], kind="stable")
        t.append(t_col[sel][o])
        y.append(y_col[sel][o])
        off.append(off[-1] + sel.size)
    return ids, np.concatenate(t), np.concatenate(y), np.asarray(off)


def test_dataset_grouping_equals_per_individual_scans():
    from magma_amd.engine import Dataset
    rng = np.random.default_rng(5)
    sizes = [7, 1, 30, 12, 64, 3]
    names = ["b10", "a2", "10", "9", "zz", "b1"]  # first-appearance order != sorted order
    ids = np.concatenate([[nm] * n for nm, n in zip(names, sizes)])
    t = np.concatenate([rng.choice(np.arange(200) * 0.05, n, replace=False) for n in sizes])
    y = rng.normal(size=t.size)
    # interleave the rows of different individuals, keeping each ID's first appearance order
    perm = np.argsort(rng.random(t.size) + np.repeat(np.arange(len(sizes)), sizes) * 0.3, kind="stable")
    db = {"ID": ids[perm], "Timestamp": t[perm], "Output": y[perm]}
    ds = Dataset(db, extra_timestamps=[11.0, 0.05])
    ids_l, t_l, y_l, off_l = _loop_grouping(db)
    assert [str(i) for i in ds.ids] == [str(i) for i in ids_l]
    assert np.array_equal(ds.off, off_l) and np.array_equal(ds.t, t_l) and np.array_equal(ds.y, y_l)
    assert np.array_equal(ds.grid, np.unique(np.concatenate([t, [11.0, 0.05]])))
    assert np.array_equal(ds.grid[ds.idx], ds.t) and ds.nmax == 64
    with pytest.raises(ValueError, match="duplicated"):
        Dataset({"ID": ["a", "b", "a"], "Timestamp": [1.0, 1.0, 1.0], "Output": [0.0, 0.0, 0.0]})
    big = Dataset({"ID": ["a"] * 65, "Timestamp": np.arange(65.0), "Output": np.zeros(65)})
    assert big.nmax == 65  # (such individuals take the native driver's dense route)
    empty = Dataset({"ID": [], "Timestamp": [], "Output": []})
    assert empty.M == 0 and empty.N == 0


def test_dataset_grouping_is_one_pass_at_full_size():
    """C3's db as a long table (5e5 rows, 1e4 string IDs): seconds, not an O(M R) scan."""
    import time
    from magma_amd.engine import Dataset
    rng = np.random.default_rng(6)
    M, n = 10000, 50
    ids = np.repeat(np.array([f"id{k}" for k in range(M)]), n)
    t = (rng.integers(0, 2000, size=(M, 1)) + np.arange(n)[None, :] * 7) % 2000 * 0.005
    t = t.reshape(-1)
    perm = rng.permutation(M * n)
    t0 = time.perf_counter()
    ds = Dataset({"ID": ids[perm], "Timestamp": t[perm], "Output": np.zeros(M * n)})
    dt = time.perf_counter() - t0
    assert ds.M == M and ds.nmax == n and np.all(np.diff(ds.off) == n)
    assert dt < 20.0, dt


def test_read_csv2_reads_the_reference_format(tmp_path):
    from magma_amd.io import read_csv2, split_by
    p = tmp_path / "db.csv"
    p.write_text("ID;Timestamp;Output;a;b;sigma;ID_dataset\n"
                 "1; 0,05; 1,091629e+01;0,31;0,86;0,79;  1\n"
                 "1; 0,20;-1,035429e+01;0,31;0,86;0,79;  1\n"
                 "12;10,00; 2,5e-03;1,5;0;0,79;  2\n")
    tb = read_csv2(str(p))
    assert list(tb) == ["ID", "Timestamp", "Output", "a", "b", "sigma", "ID_dataset"]
    assert list(tb["ID"]) == ["1", "1", "12"] and list(tb["ID_dataset"]) == ["1", "1", "2"]
    assert np.array_equal(tb["Timestamp"], [0.05, 0.2, 10.0])
    assert np.array_equal(tb["Output"], [10.91629, -10.35429, 0.0025])
    parts = split_by(tb)
    assert list(parts) == ["1", "2"] and parts["1"]["Output"].size == 2
    ref = "/root/reference/Simulations/Data/db_rep_100_M_20_N_30_time_FALSE_hp_TRUE.csv"
    if os.path.exists(ref):  # the reference's own file against the fixture decoded from it
        gold = np.load(os.path.join(ROOT, "tests", "golden", "data_FT.npz"))
        tb = read_csv2(ref)
        assert np.array_equal(tb["ID"].astype(np.int64), gold["ID"])
        assert np.array_equal(tb["ID_dataset"].astype(np.int64), gold["ID_dataset"])
        assert np.array_equal(tb["Timestamp"], gold["Timestamp"])
        assert np.array_equal(tb["Output"], gold["Output"])
