"""GPU parity tests: every HIP entry point against the CPU oracle on the same inputs.

Tolerances (FP64, north star: 1e-8 relative): kernel builds 1e-13; inverses / log-dets 1e-9
relative to the matrix scale (condition numbers up to ~1e6 in these cases); objective values
1e-10; gradients 1e-8.
"""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import magma_oracle as orc  # noqa: E402  (checker only)
from kat_common import dataset, hp_of  # noqa: E402


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import magma_amd
    from magma_amd import _lib
    return {"torch": torch, "lib": _lib.load(), "_lib": _lib, "m": magma_amd}


def _stream(torch):
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr())


def _ragged_db(rng, sizes, tmax=10.0, step=0.05):
    grid = np.arange(0.0, tmax + 1e-9, step)
    ids, ts, ys = [], [], []
    for k, n in enumerate(sizes):
        t = np.sort(rng.choice(grid, n, replace=False))
        ids += [k + 1] * n
        ts.append(t)
        ys.append(rng.normal(0, 2, n) + 0.5 * t)
    return {"ID": np.array(ids), "Timestamp": np.concatenate(ts), "Output": np.concatenate(ys)}


def _random_hp(rng, ids):
    return {"theta_0": np.array([rng.uniform(0, 3), rng.uniform(0, 2)]),
            "theta_i": {i: np.array([rng.uniform(-1, 4), rng.uniform(-0.5, 2), rng.uniform(0.1, 2)])
                        for i in ids}}


# ---------------------------------------------------------------- MFMA tile / dense kernels
def test_gemm_nt_matches_numpy(gpu):
    """Asymmetric operands: catches a swapped accumulator row/column map of the f64 MFMA."""
    torch, _lib = gpu["torch"], gpu["_lib"]
    rng = np.random.default_rng(0)
    Mr, Nc, K = 128, 192, 80
    A = rng.normal(size=(Mr, K))
    B = rng.normal(size=(Nc, K))
    C0 = rng.normal(size=(Mr, Nc))
    dA = torch.from_numpy(A.T.copy()).cuda()  # column-major
    dB = torch.from_numpy(B.T.copy()).cuda()
    dC = torch.from_numpy(C0.T.copy()).cuda()
    _lib.call("magma_gemm_nt", Mr, Nc, K, 1.5, _p(dA), Mr, _p(dB), Nc, -0.5, _p(dC), Mr,
              _stream(torch))
    got = dC.cpu().numpy().T
    exp = 1.5 * A @ B.T - 0.5 * C0
    assert np.abs(got - exp).max() < 1e-12 * np.abs(exp).max()


# dag_workgroups: 256 = the one-launch tile-graph inverse (default), 3 = the same with fewer
# workgroups than tasks in flight (its ticket order must not need co-residency), 0 = the
# launch-per-step path it replaced.
# dag_walker: who walks the chain of diagonal tiles: 2 = one workgroup whose wavefronts work a step
# ahead of the factor (default), 1 = one workgroup in lock step, 0 = a task per block step.
@pytest.fixture(params=[(256, 2), (3, 2), (256, 1), (3, 1), (256, 0), (0, 2)],
                ids=["dag256", "dag3", "dag256-walker1", "dag3-walker1", "dag256-tasks", "launches"])
def inverse_path(request, gpu):
    wgs, walker = request.param
    gpu["_lib"].call("magma_set_option", b"dag_workgroups", C.c_long(wgs))
    gpu["_lib"].call("magma_set_option", b"dag_walker", C.c_long(walker))
    yield wgs
    gpu["_lib"].call("magma_set_option", b"dag_workgroups", C.c_long(256))
    gpu["_lib"].call("magma_set_option", b"dag_walker", C.c_long(2))


@pytest.mark.parametrize("N", [1, 37, 64, 65, 200, 450, 1100])
def test_spd_inverse_and_logdet(gpu, inverse_path, N):
    torch, lib, _lib = gpu["torch"], gpu["lib"], gpu["_lib"]
    rng = np.random.default_rng(N)
    X = rng.normal(size=(N, N + 5))
    A = X @ X.T / N + 0.1 * np.eye(N)
    Np = lib.magma_pad(N)
    dA = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
    src = torch.from_numpy(A.T.copy()).cuda()
    st = _stream(torch)
    _lib.call("magma_copy_pad", N, _p(src), N, _p(dA), 0, st)
    work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
    ld = torch.zeros(1, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.call("magma_spd_inverse", N, _p(dA), _p(ld), _p(info), _p(work), st)
    out = torch.empty(N * N, dtype=torch.float64, device="cuda")
    _lib.call("magma_copy_unpad", N, _p(dA), _p(out), N, st)
    got = out.cpu().numpy().reshape(N, N).T
    exp = np.linalg.inv(A)
    assert int(info.item()) == 0
    assert np.abs(got - exp).max() < 1e-10 * np.abs(exp).max()
    assert abs(ld.item() - np.linalg.slogdet(A)[1]) < 1e-10 * max(1.0, abs(ld.item()))
    assert np.abs(got - got.T).max() == 0.0  # mirrored store: exactly symmetric


def test_tile_graph_inverse_beyond_64_block_rows(gpu):
    """Round 3: the one-launch tile graph is the default at every size (it used to hand over to the
    launch-per-step path above 64 block rows). N = 4700 = 74 block rows: against that path (same
    Cholesky-based algorithm, other summation order) and against A A^-1 = I."""
    torch, lib, _lib = gpu["torch"], gpu["lib"], gpu["_lib"]
    N = 4700
    Np = lib.magma_pad(N)
    st = _stream(torch)
    g = torch.zeros(Np, dtype=torch.float64, device="cuda")
    g[:N] = torch.linspace(0, 10, N, dtype=torch.float64)
    A0 = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
    _lib.call("magma_kern_to_cov_grid", N, _p(g), 1.0, 1.0, 0.5, _p(A0), st)
    work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
    res = {}
    try:
        for name, wgs in (("graph", 256), ("launches", 0)):
            _lib.call("magma_set_option", b"dag_workgroups", C.c_long(wgs))
            A = A0.clone()
            ld = torch.zeros(1, dtype=torch.float64, device="cuda")
            info = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.call("magma_spd_inverse", N, _p(A), _p(ld), _p(info), _p(work), st)
            torch.cuda.synchronize()
            assert int(info.item()) == 0
            res[name] = (A.view(Np, Np)[:N, :N].clone(), ld.item())
    finally:
        _lib.call("magma_set_option", b"dag_workgroups", C.c_long(256))
    Ag, ldg = res["graph"]
    Al, ldl = res["launches"]
    K = A0.view(Np, Np)[:N, :N]
    x = torch.randn(N, 4, dtype=torch.float64, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    assert (Ag @ (K @ x) - x).abs().max().item() < 1e-9
    assert (Ag - Al).abs().max().item() < 1e-9 * Ag.abs().max().item()  # cond(K) ~ 1e4
    assert abs(ldg - ldl) < 1e-9 * abs(ldg)
    assert (Ag - Ag.T).abs().max().item() == 0.0


@pytest.mark.parametrize("N,bad", [(70, 40), (300, 130)])
def test_spd_inverse_flags_indefinite(gpu, inverse_path, N, bad):
    torch, lib, _lib = gpu["torch"], gpu["lib"], gpu["_lib"]
    A = np.eye(N) + 0.01 * np.ones((N, N))
    A[bad, bad] = -1.0
    Np = lib.magma_pad(N)
    dA = torch.empty(Np * Np, dtype=torch.float64, device="cuda")
    src = torch.from_numpy(A.T.copy()).cuda()
    st = _stream(torch)
    _lib.call("magma_copy_pad", N, _p(src), N, _p(dA), 0, st)
    work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
    ld = torch.zeros(1, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.call("magma_spd_inverse", N, _p(dA), _p(ld), _p(info), _p(work), st)
    assert int(info.item()) == bad + 1


# ---------------------------------------------------------------- covariance builders
def test_kern_to_cov_and_inv_vector_branch(gpu):
    m = gpu["m"]
    rng = np.random.default_rng(3)
    t = rng.choice(np.arange(0, 10, 0.05), 150, replace=False)  # unsorted on purpose
    lab, K = m.kern_to_cov(t, m.kernel, [1.3, 0.7], 0.4)
    lab_o, K_o = orc.kern_to_cov(t, orc.kernel, [1.3, 0.7], 0.4)
    assert np.array_equal(lab, lab_o)
    assert np.abs(K - K_o).max() < 1e-13 * np.abs(K_o).max()
    lab, Ki = m.kern_to_inv(t, m.kernel_mu, [1.3, 0.7], 0.4)
    _, Ki_o = orc.kern_to_inv(t, orc.kernel, [1.3, 0.7], 0.4)
    assert np.abs(Ki - Ki_o).max() < 1e-9 * np.abs(Ki_o).max()


@pytest.mark.parametrize("sizes", [[1, 2, 3, 7, 8, 9, 30, 31, 33, 50, 64], [30] * 5])
def test_kern_to_cov_and_inv_batched(gpu, sizes):
    m = gpu["m"]
    rng = np.random.default_rng(len(sizes))
    db = _ragged_db(rng, sizes)
    hp = _random_hp(rng, orc.unique_in_order(db["ID"]))
    got = m.kern_to_cov(db, m.kernel, hp["theta_i"], 0)
    exp = orc.kern_to_cov(db, orc.kernel, hp["theta_i"], 0)
    goti = m.kern_to_inv(db, m.kernel, hp["theta_i"], 0)
    expi = orc.kern_to_inv(db, orc.kernel, hp["theta_i"], 0)
    for i in exp:
        assert np.array_equal(got[i][0], exp[i][0])
        assert np.abs(got[i][1] - exp[i][1]).max() < 1e-13 * np.abs(exp[i][1]).max()
        assert np.abs(goti[i][1] - expi[i][1]).max() < 1e-9 * np.abs(expi[i][1]).max(), i
    # common numeric theta + sigma (CF:89-93)
    got = m.kern_to_inv(db, m.kernel, [0.5, 1.0], 0.3)
    exp = orc.kern_to_inv(db, orc.kernel, [0.5, 1.0], 0.3)
    for i in exp:
        assert np.abs(got[i][1] - exp[i][1]).max() < 1e-9 * np.abs(exp[i][1]).max()


# ---------------------------------------------------------------- E-step
@pytest.mark.parametrize("sizes", [[30] * 20, [1, 5, 17, 40, 64, 12, 3]])
def test_e_step_matches_oracle(gpu, sizes):
    m = gpu["m"]
    rng = np.random.default_rng(sum(sizes))
    db = _ragged_db(rng, sizes)
    hp = _random_hp(rng, orc.unique_in_order(db["ID"]))
    got = m.e_step(db, 1.5, m.kernel_mu, m.kernel, hp)
    exp = orc.e_step(db, 1.5, orc.kernel, orc.kernel, hp)
    assert np.array_equal(got["mean"]["Timestamp"], exp["mean"]["Timestamp"])
    sc = np.abs(exp["cov"][1]).max()
    assert np.abs(got["cov"][1] - exp["cov"][1]).max() < 1e-9 * sc
    assert np.abs(got["mean"]["Output"] - exp["mean"]["Output"]).max() < 1e-9 * np.abs(exp["mean"]["Output"]).max()
    assert np.abs(got["pred_GP"]["Var"] - exp["pred_GP"]["Var"]).max() < 1e-9 * sc


def test_posterior_mu_vector_prior_mean(gpu):
    m = gpu["m"]
    truth, test, train = dataset("data_FF", 3)
    hp = hp_of("hp_FF", 3)
    got = m.posterior_mu(train, test, truth["Timestamp"][::7], 0.3, m.kernel_mu, m.kernel, hp)
    exp = orc.posterior_mu(train, test, truth["Timestamp"][::7], 0.3, orc.kernel, orc.kernel, hp)
    assert np.abs(got["cov"][1] - exp["cov"][1]).max() < 1e-9 * np.abs(exp["cov"][1]).max()
    assert np.abs(got["mean"]["Output"] - exp["mean"]["Output"]).max() < 1e-9


# ---------------------------------------------------------------- M-step objective / gradient
def _posterior_for(db, hp, m0=0.7):
    return orc.e_step(db, m0, orc.kernel, orc.kernel, hp)


@pytest.mark.parametrize("sizes", [[1, 2, 12, 30, 50, 64], [30] * 4])
def test_logL_GP_mod_and_gradient_per_individual(gpu, sizes):
    m = gpu["m"]
    rng = np.random.default_rng(7 + len(sizes))
    db = _ragged_db(rng, sizes)
    ids = orc.unique_in_order(db["ID"])
    hp = _random_hp(rng, ids)
    par = _posterior_for(db, hp)
    labels, cov = par["cov"]
    for i in ids:
        sel = db["ID"] == i
        db_i = {"Timestamp": db["Timestamp"][sel], "Output": db["Output"][sel]}
        mean_i = orc._mean_at(par["mean"], db_i["Timestamp"])
        cov_i = orc._block(labels, cov, db_i["Timestamp"])
        th = hp["theta_i"][i] + rng.normal(0, 0.1, 3)
        f = m.logL_GP_mod(th, db_i, mean_i, m.kernel, cov_i)
        g = m.gr_GP_mod(th, db_i, mean_i, m.kernel, cov_i)
        f_o = orc.logL_GP_mod(th, db_i, mean_i, orc.kernel, cov_i)
        g_o = orc.gr_GP_mod(th, db_i, mean_i, orc.kernel, cov_i)
        assert abs(f - f_o) < 1e-10 * max(1.0, abs(f_o)), (i, f, f_o)
        assert np.abs(g - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max()), (i, g, g_o)
        # plain GP likelihood / gradient (train_new_gp's objective)
        f = m.logL_GP(th, db_i, mean_i, m.kernel, cov_i)
        g = m.gr_GP(th, db_i, mean_i, m.kernel, cov_i)
        f_o = orc.logL_GP(th, db_i, mean_i, orc.kernel, cov_i)
        g_o = orc.gr_GP(th, db_i, mean_i, orc.kernel, cov_i)
        assert abs(f - f_o) < 1e-10 * max(1.0, abs(f_o))
        assert np.abs(g - g_o).max() < 1e-8 * max(1.0, np.abs(g_o).max())


def test_theta0_objective_and_gradient(gpu):
    m = gpu["m"]
    rng = np.random.default_rng(11)
    db = _ragged_db(rng, [30] * 10)
    hp = _random_hp(rng, orc.unique_in_order(db["ID"]))
    par = _posterior_for(db, hp)
    th0 = np.array([1.2, 0.4])
    f = m.logL_GP_mod(th0, par["mean"], 0.7, m.kernel_mu, par["cov"][1], pen_diag=0.8)
    g = m.gr_GP_mod(th0, par["mean"], 0.7, m.kernel_mu, par["cov"][1], pen_diag=0.8)
    f_o = orc.logL_GP_mod(th0, par["mean"], 0.7, orc.kernel, par["cov"][1], pen_diag=0.8)
    g_o = orc.gr_GP_mod(th0, par["mean"], np.full(par["mean"]["Output"].size, 0.7), orc.kernel,
                        par["cov"][1], pen_diag=0.8)
    assert abs(f - f_o) < 1e-10 * abs(f_o), (f, f_o)
    assert np.abs(g - g_o).max() < 1e-8 * np.abs(g_o).max(), (g, g_o)


def test_common_hp_objective_and_monitoring(gpu):
    m = gpu["m"]
    rng = np.random.default_rng(13)
    db = _ragged_db(rng, [20, 25, 30, 1, 8])
    ids = orc.unique_in_order(db["ID"])
    hp = _random_hp(rng, ids)
    par = _posterior_for(db, hp)
    th = np.array([0.8, 0.9, 0.5])
    f = m.logL_GP_mod_common_hp(th, db, par["mean"], m.kernel, par["cov"])
    g = m.gr_GP_mod_common_hp(th, db, par["mean"], m.kernel, par["cov"])
    f_o = orc.logL_GP_mod_common_hp(th, db, par["mean"], orc.kernel, par["cov"])
    g_o = orc.gr_GP_mod_common_hp(th, db, par["mean"], orc.kernel, par["cov"])
    assert abs(f - f_o) < 1e-10 * abs(f_o)
    assert np.abs(g - g_o).max() < 1e-8 * np.abs(g_o).max()
    ll = m.logL_monitoring(hp, db, m.kernel, m.kernel_mu, par["mean"], par["cov"], 0.7)
    ll_o = orc.logL_monitoring(hp, db, orc.kernel, orc.kernel, par["mean"], par["cov"], 0.7)
    assert abs(ll - ll_o) < 1e-10 * abs(ll_o)


# ---------------------------------------------------------------- prediction
def test_pred_gp_matches_oracle(gpu):
    m = gpu["m"]
    truth, test, train = dataset("data_FT", 5)
    hp = hp_of("hp_FT", 5)
    db_obs = {k: v[:20] for k, v in test.items()}
    ts = test["Timestamp"][20:]
    mu = orc.posterior_mu(train, db_obs, ts, 0.0, orc.kernel, orc.kernel, hp)
    got = m.pred_gp(db_obs, ts, mu["mean"], mu["cov"], m.kernel, [1.1, 0.6], 0.3)
    exp = orc.pred_gp(db_obs, ts, mu["mean"], mu["cov"], orc.kernel, [1.1, 0.6], 0.3)
    assert np.abs(got["Mean"] - exp["Mean"]).max() < 1e-9 * np.abs(exp["Mean"]).max()
    assert np.abs(got["Var"] - exp["Var"]).max() < 1e-9 * np.abs(exp["Var"]).max()
    # plain GP regression: cov_mu NULL, constant mean, default 500-point grid (MG:152-160)
    got = m.pred_gp(db_obs, None, 2.0, None, m.kernel, [1.1, 0.6], 0.3)
    exp = orc.pred_gp(db_obs, None, 2.0, None, orc.kernel, [1.1, 0.6], 0.3)
    assert np.abs(got["Mean"] - exp["Mean"]).max() < 1e-9 * np.abs(exp["Mean"]).max()
    assert np.abs(got["Var"] - exp["Var"]).max() < 1e-9 * np.abs(exp["Var"]).max()


def test_em_iteration_shortcuts_equal_explicit_monitoring(gpu):
    """The engine takes logL_monitoring(hp) from by-products (K_0^-1 of the E-step, the
    optimisers' first evaluations) and reuses K_0^-1 between monitoring(new hp) and the next
    E-step: both must equal the explicit computations and the oracle."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(21)
    db = _ragged_db(rng, [30, 12, 40, 7, 25, 33])
    ds = Dataset(db)
    eng = Engine(ds)
    th0 = np.array([1.0, 1.0])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    for common in (False, True):
        n0, ni, l_old, l_new = eng.em_iteration(th0, thi, 0.4, common)
        hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
        par = orc.e_step(db, 0.4, orc.kernel, orc.kernel, hp)
        ll = orc.logL_monitoring(hp, db, orc.kernel, orc.kernel, par["mean"], par["cov"], 0.4)
        assert abs(l_old - ll) < 1e-9 * abs(ll), (common, l_old, ll)
        assert abs(eng.logL_monitoring(th0, thi) - ll) < 1e-9 * abs(ll)
        hp_n = {"theta_0": n0, "theta_i": {i: ni[k] for k, i in enumerate(ds.ids)}}
        ll_n = orc.logL_monitoring(hp_n, db, orc.kernel, orc.kernel, par["mean"], par["cov"], 0.4)
        assert abs(l_new - ll_n) < 1e-9 * abs(ll_n)
        # next E-step right after monitoring(new hp): K_0^-1 reuse path vs a cold engine
        eng.logL_monitoring(n0, ni)
        eng.e_step(n0, ni, float(np.mean(ni[:, 2])), 0.4)
        cold = Engine(ds)
        cold.e_step(n0, ni, float(np.mean(ni[:, 2])), 0.4)
        # (to rounding: the K_0^-1 reuse path sums P + K_0^-1 from a different K_0^-1 launch)
        c1, c2 = eng.be.get_cov(), cold.be.get_cov()
        assert np.abs(c1 - c2).max() < 1e-12 * np.abs(c2).max()
        assert abs(eng.last_f0 - cold.last_f0) <= 1e-11 * abs(cold.last_f0)


def test_native_mstep_loop_equals_python_driven_loop(gpu):
    """magma_indiv_mstep (C++ round loop, compact ask/tell) against minimize_batch driving
    magma_indiv_eval from Python: same optimiser, same evaluations -> identical results."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(23)
    db = _ragged_db(rng, [5, 30, 17, 50, 42, 30, 8, 64])
    ids = orc.unique_in_order(db["ID"])
    hp = _random_hp(rng, ids)
    th0 = hp["theta_0"]
    thi = np.array([hp["theta_i"][i] for i in ids])
    ds = Dataset(db)
    res = []
    eng = Engine(ds)  # one E-step for both routes
    eng.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.3)
    for native in (True, False):
        eng.native_mstep = native
        eng.stats["indiv_rounds"] = 0
        n0, ni = eng.m_step(th0, thi, False)
        res.append((n0, ni, eng.last_sum_fi, eng.stats["indiv_rounds"]))
    assert np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]
    # and against the oracle's m_step with scipy's L-BFGS-B
    par = orc.e_step(db, 0.3, orc.kernel, orc.kernel, hp)
    new = orc.m_step(db, hp, par["mean"], par["cov"], orc.kernel, orc.kernel, 0.3, False)
    exp = np.array([new["theta_i"][i] for i in ids])
    assert np.abs(res[0][1] - exp).max() < 1e-6 * max(1.0, np.abs(exp).max())


def test_resident_mstep_equals_lock_step_mstep(gpu):
    """magma_indiv_mstep_resident (L-BFGS-B stepped inside the evaluating wavefronts) against the
    host-driven lock-step loop: same optimiser source, same evaluation code -> same iterates."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(29)
    db = _ragged_db(rng, [5, 30, 17, 50, 42, 30, 8, 64, 1, 2])
    ids = orc.unique_in_order(db["ID"])
    hp = _random_hp(rng, ids)
    th0 = hp["theta_0"]
    thi = np.array([hp["theta_i"][i] for i in ids])
    eng = Engine(Dataset(db))
    eng.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.3)
    a = eng.be.indiv_mstep_resident(thi)
    b = eng.be.indiv_mstep(thi)
    assert np.array_equal(a["status"], b["status"]), (a["status"], b["status"])
    assert np.array_equal(a["nfev"], b["nfev"]) and np.array_equal(a["iters"], b["iters"])
    assert np.abs(a["x"] - b["x"]).max() <= 1e-12 * max(1.0, np.abs(b["x"]).max())
    assert np.abs(a["f_x0"] - b["f_x0"]).max() <= 1e-12 * np.abs(b["f_x0"]).max()
    assert np.abs(a["f"] - b["f"]).max() <= 1e-12 * np.abs(b["f"]).max()


@pytest.mark.parametrize("sizes", [[5, 30, 17, 50, 42, 30, 8, 48, 1, 2], [50] * 37, [40], [33, 12],
                                   [12, 9, 16], [20] * 5, [56, 3]],
                         ids=["ragged", "37x50", "one", "two", "one-tile", "two-tiles", "56-rows"])
def test_resident_mstep_helper_wavefronts_do_not_change_results(gpu, sizes):
    """Idle wavefronts of a workgroup take column blocks of the evaluations still running next to
    them (magma_set_option("resident_help", ...): 0 off, 1 on, 3 on with every compute unit
    launched, so that most workgroups consist of helpers only). Block sums are added in block
    order whoever computed them: every mode gives the bits of the lock-step loop, and no workgroup
    gave up on a wait (magma_debug_resident_diag word 0)."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    _lib = gpu["_lib"]
    rng = np.random.default_rng(31)
    db = _ragged_db(rng, sizes)
    ids = orc.unique_in_order(db["ID"])
    hp = _random_hp(rng, ids)
    thi = np.array([hp["theta_i"][i] for i in ids])
    eng = Engine(Dataset(db))
    eng.e_step(hp["theta_0"], thi, float(np.mean(thi[:, 2])), 0.3)
    ref = eng.be.indiv_mstep(thi)
    try:
        for mode in (0, 1, 3, 3):
            _lib.call("magma_set_option", b"resident_help", C.c_long(mode))
            a = eng.be.indiv_mstep_resident(thi)
            for k in ("status", "nfev", "iters", "x", "f", "f_x0"):
                assert np.array_equal(a[k], ref[k], equal_nan=True), (mode, k)
    finally:
        _lib.call("magma_set_option", b"resident_help", C.c_long(1))
    diag = (C.c_ulonglong * 24)()
    _lib.call("magma_debug_resident_diag", diag)
    assert diag[0] == 0, list(diag)


def test_results_do_not_depend_on_stale_lds(gpu):
    """LDS is not cleared between workgroups. Every kernel's padding tricks (identity padding,
    fragment reads that run past a row, zero-padded operands) must select stale words away, never
    multiply them by zero: with the LDS of every CU filled with NaN patterns first, the parity
    tests above must still pass. (Found as a sporadic NaN of the f-only evaluation.)"""
    _lib = gpu["_lib"]

    def poison():
        _lib.call("magma_debug_poison_lds", _stream(gpu["torch"]))
    cases = [
        (test_kern_to_cov_and_inv_batched, ([1, 2, 3, 7, 8, 9, 30, 31, 33, 50, 64],)),
        (test_e_step_matches_oracle, ([1, 5, 17, 40, 64, 12, 3],)),
        (test_logL_GP_mod_and_gradient_per_individual, ([1, 2, 12, 30, 50, 64],)),
        (test_theta0_objective_and_gradient, ()),
        (test_common_hp_objective_and_monitoring, ()),
        (test_pred_gp_matches_oracle, ()),
        (test_spd_inverse_and_logdet, (256, 200)),
        (test_em_iteration_shortcuts_equal_explicit_monitoring, ()),
        (test_resident_mstep_equals_lock_step_mstep, ()),
    ]
    for fn, args in cases:
        poison()
        fn(gpu, *args)
    # the f-only evaluation on its own, many times over (the kernel that showed it)
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(31)
    db = _ragged_db(rng, [30, 12, 40, 7, 25, 33, 50, 1])
    ds = Dataset(db)
    eng = Engine(ds)
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    eng.e_step(np.array([1.0, 1.0]), thi, 0.2, 0.4)
    ref, _ = eng.be.indiv_eval(thi, False)
    assert np.all(np.isfinite(ref))
    for _ in range(20):
        poison()
        f, _ = eng.be.indiv_eval(thi, False)
        assert np.array_equal(f, ref)


def test_pred_gp_batched_matches_oracle_and_single_calls(gpu):
    """many new individuals predicted in one launch against a device-resident
    hyper-posterior; against the oracle's posterior_mu + pred_gp and the single-call route."""
    m = gpu["m"]
    truth, test, train = dataset("data_FT", 7)
    hp = hp_of("hp_FT", 7)
    rng = np.random.default_rng(41)
    grid_t = np.arange(0.0, 10.0001, 0.05).round(2)
    new_dbs, tss, thetas, sigmas = [], [], [], []
    for b, n in enumerate([20, 1, 7, 33, 64, 12]):
        t = np.sort(rng.choice(grid_t, n, replace=False))
        perm = rng.permutation(n)  # unsorted input rows, as a db may come
        new_dbs.append({"Timestamp": t[perm], "Output": (np.sin(t) * 3 + rng.normal(size=n))[perm]})
        tss.append(np.sort(rng.choice(grid_t, [5, 17, 0, 40, 9, 1][b], replace=False)))
        thetas.append([rng.uniform(0, 2), rng.uniform(0, 1)])
        sigmas.append(rng.uniform(0.1, 1.0))
    eng = m.posterior_mu_resident(train, new_dbs, np.concatenate(tss), 0.0, m.kernel_mu, m.kernel, hp)
    got = m.pred_gp_batched(eng, new_dbs, tss, np.array(thetas), np.array(sigmas))
    all_new = {"Timestamp": np.concatenate([d["Timestamp"] for d in new_dbs])}
    mu = orc.posterior_mu(train, all_new, np.concatenate(tss), 0.0, orc.kernel, orc.kernel, hp)
    for b, nd in enumerate(new_dbs):
        assert got[b]["Mean"].size == tss[b].size
        if tss[b].size == 0:
            continue
        o = np.argsort(nd["Timestamp"])
        db_b = {k: v[o] for k, v in nd.items()}
        exp = orc.pred_gp(db_b, tss[b], mu["mean"], mu["cov"], orc.kernel, thetas[b], sigmas[b])
        sc_m, sc_v = np.abs(exp["Mean"]).max(), np.abs(exp["Var"]).max()
        assert np.abs(got[b]["Mean"] - exp["Mean"]).max() < 1e-8 * max(1.0, sc_m), b
        assert np.abs(got[b]["Var"] - exp["Var"]).max() < 1e-8 * max(1.0, sc_v), b
        one = m.pred_gp(db_b, tss[b], mu["mean"], mu["cov"], m.kernel, thetas[b], sigmas[b])
        assert np.abs(got[b]["Mean"] - one["Mean"]).max() < 1e-8 * max(1.0, sc_m)
        assert np.abs(got[b]["Var"] - one["Var"]).max() < 1e-8 * max(1.0, sc_v)
    # shared targets and shared hyper-parameters
    got2 = m.pred_gp_batched(eng, new_dbs[:2], tss[0], thetas[0], sigmas[0])
    assert np.array_equal(got2[0]["Mean"], got[0]["Mean"])
    with pytest.raises(KeyError):
        m.pred_gp_batched(eng, new_dbs[:1], np.array([0.123]), thetas[0], sigmas[0])


@pytest.mark.parametrize("N", [130, 700, 2000])
def test_spd_inverse_is_bitwise_repeatable_next_to_other_work(gpu, N):
    """The one-launch inverse hands tiles between workgroups inside the launch; whichever
    workgroup takes which task and whatever else occupies the chip, every tile is summed in a
    fixed order: repeated launches, alone and next to a busy second stream, must agree bit for
    bit (and never abandon: info == 0)."""
    torch, lib, _lib = gpu["torch"], gpu["lib"], gpu["_lib"]
    Np = lib.magma_pad(N)
    gen = torch.Generator(device="cuda").manual_seed(N)
    t = torch.sort(torch.rand(N, dtype=torch.float64, device="cuda", generator=gen) * 10)[0]
    K = torch.exp(1.0 - 0.5 * np.exp(-1.0) * (t[:, None] - t[None, :]) ** 2) \
        + 0.3 * torch.eye(N, dtype=torch.float64, device="cuda")
    A0 = torch.eye(Np, dtype=torch.float64, device="cuda")
    A0[:N, :N] = K
    A = A0.clone().reshape(-1)
    work = torch.empty((lib.magma_workspace_bytes(N) + 7) // 8, dtype=torch.float64, device="cuda")
    ld = torch.zeros(1, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    big = torch.ones(2048, 2048, dtype=torch.float64, device="cuda")
    ref = None
    for it in range(40):
        A.copy_(A0.reshape(-1))
        if it % 2 == 1:
            with torch.cuda.stream(side):
                big @ big
        _lib.call("magma_spd_inverse", N, _p(A), _p(ld), _p(info), _p(work), _stream(torch))
        torch.cuda.synchronize()
        assert int(info.item()) == 0, (it, int(info.item()))
        if ref is None:
            ref = (A.clone(), ld.clone())
        else:
            assert torch.equal(A, ref[0]) and torch.equal(ld, ref[1]), it
    resid = (K @ ref[0].view(Np, Np)[:N, :N]
             - torch.eye(N, dtype=torch.float64, device="cuda")).abs().max().item()
    assert resid < 1e-10, resid


def test_em_iteration_is_bitwise_repeatable_and_routes_agree(gpu):
    """Default E-step route ("gather": fixed summation order) makes a whole EM iteration
    repeatable bit for bit; the atomic route agrees with it to rounding, and both with the
    oracle (flagged individuals included: they contribute through the slow path, as
    tryCatch(solve, ginv) does at CF:161)."""
    from magma_amd.engine import Dataset
    from host_mirror import Engine
    rng = np.random.default_rng(43)
    db = _ragged_db(rng, [30, 12, 40, 7, 25, 33, 1, 64, 50, 50])
    ds = Dataset(db)
    th0 = np.array([0.9, 0.7])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1)) + rng.uniform(-0.3, 0.3, size=(ds.M, 3))
    runs = []
    for _ in range(2):
        eng = Engine(ds)
        assert eng.be.scatter_route == "gather"
        out = eng.em_iteration(th0, thi, 0.4, False)
        runs.append((out, eng.be.get_cov(), eng.be.get_mean()))
    (a, cov_a, mean_a), (b, cov_b, mean_b) = runs
    assert np.array_equal(cov_a, cov_b) and np.array_equal(mean_a, mean_b)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and a[3] == b[3]
    eng = Engine(ds)
    eng.be.scatter_route = "atomic"
    eng.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.4)
    cov_at = eng.be.get_cov()
    eng2 = Engine(ds)
    eng2.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.4)
    cov_g = eng2.be.get_cov()
    assert np.abs(cov_at - cov_g).max() < 1e-12 * np.abs(cov_g).max()
    # a flagged individual (sigma = 0 and two near-identical timestamps make K_i singular to
    # rounding -> info != 0) is left out by both routes
    thi_bad = thi.copy()
    thi_bad[3] = [5.0, 8.0, 0.0]
    covs = []
    for route in ("gather", "atomic"):
        e = Engine(ds)
        e.be.scatter_route = route
        e.e_step(th0, thi_bad, 0.3, 0.4)
        covs.append((e.be.get_cov(), e.be.n_flagged()))
    assert covs[0][1] == covs[1][1]
    assert np.abs(covs[0][0] - covs[1][0]).max() < 1e-12 * np.abs(covs[1][0]).max()
