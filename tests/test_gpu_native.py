"""The native EM driver behind the C ABI (magma_ctx / magma_dataset / magma_training, ...: the
reference's `training`, `e_step`, `m_step`, `logL_monitoring`, `posterior_mu`, `train_new_gp`
as ONE library call each, MAGMA.R:15-133, Computing_functions.R:236-264, 371-437), driven through
ctypes only (magma_amd.native is pointer plumbing):
  * `magma_training` reproduces the hyper-parameters R stored in train_FT.rds (1e-7) and equals
    the Python-driven engine on the same kernels;
  * one EM iteration equals the Python-driven `Engine` (bit for bit where the order of summation
    is the same) and the oracle;
  * several ranks (threads on one GPU, all-reduce through the library's callback communicator)
    equal one rank: packed all-reduce of the precision, gathered theta_i, and the theta_0
    gradient whose N^3 products are split over the ranks by block rows;
  * posterior_mu / pred_gp / train_new_gp through their native entry points against the oracle.
"""
import threading
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import magma_oracle as orc  # noqa: E402  (checker only)
from kat_common import dataset, hp_of  # noqa: E402


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import magma_amd
    return magma_amd


def _ragged_db(rng, sizes, tmax=10.0, step=0.05):
    grid = np.arange(0.0, tmax + 1e-9, step)
    ids, ts, ys = [], [], []
    for k, n in enumerate(sizes):
        t = np.sort(rng.choice(grid, n, replace=False))
        ids += [k + 1] * n
        ts.append(t)
        ys.append(rng.normal(0, 2, n) + 0.5 * t)
    return {"ID": np.array(ids), "Timestamp": np.concatenate(ts), "Output": np.concatenate(ys)}


@pytest.mark.parametrize("d", [1, 2, 4])
def test_native_training_reproduces_stored_model(m, d):
    """Only `magma_training` is called (plus the dataset / context constructors)."""
    from magma_amd.native import NativeEngine
    _, _, train = dataset("data_FT", d)
    eng = NativeEngine(m.Dataset(train))
    trace = []
    th0, thi, cv = eng.training(0.0, [1.0, 1.0], [1.0, 1.0, 0.2], common_hp=True, trace=trace)
    ref = hp_of("hp_FT", d)
    assert cv and len(trace) >= 2 and trace[-1]["eps"] < 1e-3
    assert np.abs(th0 - ref["theta_0"]).max() < 1e-7, (th0, ref["theta_0"])
    for k, i in enumerate(eng.ds.ids):
        assert np.abs(thi[k] - ref["theta_i"][int(i)]).max() < 1e-7
    # the last E-step's hyper-posterior stays resident (`param`, MG:78): equals the oracle's
    # E-step at the hyper-parameters that E-step saw (those of the previous iteration)
    assert eng.get_cov().shape == (eng.N, eng.N) and np.all(np.isfinite(eng.get_mean()))
    eng.close()


@pytest.mark.parametrize("common", [False, True], ids=["individual-hp", "common-hp"])
@pytest.mark.parametrize("sizes", [[30] * 12, [30, 12, 40, 7, 25, 33, 1, 64, 50, 50]],
                         ids=["uniform", "ragged"])
def test_native_em_iteration_equals_python_engine_and_oracle(m, sizes, common):
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(len(sizes))
    db = _ragged_db(rng, sizes)
    ds = m.Dataset(db)
    th0 = np.array([0.9, 0.7])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    if not common:  # (with common HPs `training` keeps every row equal, MG:26-27, CF:420)
        thi = thi + rng.uniform(-0.3, 0.3, size=(ds.M, 3))
    from host_mirror import Engine
    nat, py = NativeEngine(ds), Engine(ds)
    pen = float(np.mean(thi[:, 2]))
    assert nat.e_step(th0, thi, pen, 0.4) == 0 and py.e_step(th0, thi, pen, 0.4) == 0
    c_n, c_p = nat.get_cov(), py.be.get_cov()
    uniform = len(set(sizes)) == 1
    if uniform:  # same order of summation in both list builders -> same bits
        assert np.array_equal(c_n, c_p) and np.array_equal(nat.get_mean(), py.be.get_mean())
        assert nat.last_f0 == py.last_f0
    else:
        assert np.abs(c_n - c_p).max() < 1e-10 * np.abs(c_p).max()
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    exp = orc.e_step(db, 0.4, orc.kernel, orc.kernel, hp)
    assert np.abs(c_n - exp["cov"][1]).max() < 1e-9 * np.abs(exp["cov"][1]).max()
    a = nat.em_iteration(th0, thi, 0.4, common)
    b = py.em_iteration(th0, thi, 0.4, common)
    # Same kernels, same optimiser source; what differs is host arithmetic in the last bit (mean
    # sigma_i and the sums over individuals: NumPy's pairwise order against a plain loop, and for
    # common HPs a device-side sum), which the optimisers' paths amplify a little.
    dth = np.abs(a[1] - b[1])
    assert np.abs(a[0] - b[0]).max() < 1e-6 and np.median(dth) < 1e-8 and dth.max() < 1e-3
    assert abs(a[2] - b[2]) < 1e-12 * abs(b[2]) and abs(a[3] - b[3]) < 1e-8 * abs(b[3])
    ll = orc.logL_monitoring(hp, db, orc.kernel, orc.kernel, exp["mean"], exp["cov"], 0.4)
    assert abs(a[2] - ll) < 1e-9 * abs(ll)
    hp_n = {"theta_0": a[0], "theta_i": {i: a[1][k] for k, i in enumerate(ds.ids)}}
    ll_n = orc.logL_monitoring(hp_n, db, orc.kernel, orc.kernel, exp["mean"], exp["cov"], 0.4)
    assert abs(a[3] - ll_n) < 1e-9 * abs(ll_n)
    assert abs(nat.logL_monitoring(a[0], a[1]) - ll_n) < 1e-9 * abs(ll_n)
    if common:
        assert np.all(a[1] == a[1][0]) and nat.stats["common_evals"] > 0
    else:
        assert nat.stats["indiv_evals"] >= ds.M and nat.stats["theta0_evals"] >= 2
    nat.close()


def test_native_training_individual_hp_equals_python_engine(m):
    from magma_amd.native import NativeEngine
    _, _, train = dataset("data_FF", 1)
    ds = m.Dataset(train)
    from host_mirror import Engine
    nat, py = NativeEngine(ds), Engine(ds)
    t_n, t_p = [], []
    a = nat.training(0.0, [1.0, 1.0], [1.0, 1.0, 0.2], common_hp=False, trace=t_n)
    b = py.training(0.0, [1.0, 1.0], [1.0, 1.0, 0.2], common_hp=False, trace=t_p)
    assert a[2] == b[2] and len(t_n) == len(t_p)
    # (last-bit differences of the host sums, and the theta_0 gradient formed through the factor
    # of cov in the driver but through the full product in the Python engine, amplified over ~6 EM
    # iterations of optimisations that stop at factr = 1e7)
    assert np.abs(a[0] - b[0]).max() < 1e-5 and np.abs(a[1] - b[1]).max() < 1e-4
    assert abs(t_n[0]["logLL"] - t_p[0]["logLL"]) <= 1e-9 * abs(t_p[0]["logLL"])
    for x, y in zip(t_n, t_p):  # (later iterations start from hyper-parameters that differ by ~1e-6)
        assert abs(x["logLL"] - y["logLL"]) <= 1e-6 * abs(y["logLL"])
    ref = hp_of("hp_FF", 1)  # at optimiser tolerance (see test_gpu_kat)
    assert np.abs(a[0] - ref["theta_0"]).max() < 1e-3
    nat.close()


# ------------------------------------------------------------------ several ranks on one GPU
class _Shared:
    def __init__(self, world):
        self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)


def _make_allreduce(shared, rank):
    from magma_amd.native import device_tensor

    def allreduce(ptr, count):
        import torch
        t = device_tensor(ptr, count)
        shared.slots[rank] = t.clone()
        torch.cuda.synchronize()
        shared.barrier.wait()
        parts = list(shared.slots)
        t.zero_()
        for p in parts:  # rank order on every rank: identical bits everywhere
            t.add_(p)
        torch.cuda.synchronize()
        shared.barrier.wait()
    return allreduce


def _run_ranks(ds, world, fn):
    import torch
    from magma_amd.native import NativeEngine
    shared = _Shared(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            eng = NativeEngine(ds, rank=r, world=world, allreduce=_make_allreduce(shared, r))
            out[r] = fn(eng)
            eng.close()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            shared.barrier.abort()
    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("world,N", [(2, 200), (3, 450), (4, 2000)])
def test_native_ranks_as_threads_equal_one_rank(m, world, N):
    """Callback communicator: the sharded E-step (lower tiles of P packed for the all-reduce), the
    theta_0 M-step with its gradient products split by block rows, the gathered theta_i."""
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_arrays
    M = 403  # not a multiple of any world size here: ragged shards, one possibly short
    a = simu_arrays(M, 30, N, seed=77)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    th0 = np.array([1.0, 1.0])
    thi = np.tile([1.0, 1.0, 0.2], (M, 1))
    one = NativeEngine(ds)
    assert one.e_step(th0, thi, 0.2, 0.3) == 0
    post = (one.get_mean(), one.get_cov())

    def job(eng):
        info = eng.e_step(th0, thi, 0.2, 0.3)
        out = dict(info=info, mean=eng.get_mean(), cov=eng.get_cov(), f0=eng.last_f0)
        eng.set_mean_cov(*post)  # bit-identical hyper-posterior for the M-step comparison
        out["n0"], out["ni"] = eng.m_step(th0, thi, False)
        out["mon"] = eng.logL_monitoring(out["n0"], out["ni"])
        out["n0c"], out["nic"] = eng.m_step(th0, thi, True)
        out["em"] = eng.em_iteration(th0, thi, 0.3, False)
        return out
    ref = job(one)
    res = _run_ranks(ds, world, job)
    rel = lambda x, y: float(np.abs(np.asarray(x) - np.asarray(y)).max() / np.abs(np.asarray(y)).max())  # noqa: E731
    for r in res:
        assert r["info"] == 0
        assert rel(r["cov"], ref["cov"]) < 1e-12 and rel(r["mean"], ref["mean"]) < 1e-12
        assert rel(r["f0"], ref["f0"]) < 1e-12
        for k in ("n0", "ni", "n0c", "nic"):  # replicated results are the same bits on every rank
            assert np.array_equal(r[k], res[0][k]), k
        assert r["mon"] == res[0]["mon"] and r["em"][2] == res[0]["em"][2]
    r0 = res[0]
    # theta_i: every individual's problem has the same inputs whichever rank runs it
    assert np.array_equal(r0["ni"], ref["ni"])
    # theta_0: same objective; the two gradient sums are added in another order (block rows per
    # rank, then ranks) -> rounding-level differences, amplified a little by the optimiser
    assert np.abs(r0["n0"] - ref["n0"]).max() < 1e-6, (r0["n0"], ref["n0"])
    assert rel(r0["mon"], ref["mon"]) < 1e-9
    assert np.abs(r0["n0c"] - ref["n0c"]).max() < 1e-6 and np.abs(r0["nic"] - ref["nic"]).max() < 1e-5
    # whole iteration: hyper-posteriors that differ by rounding, then ~400 optimisations
    assert rel(r0["em"][2], ref["em"][2]) < 1e-12 and rel(r0["em"][3], ref["em"][3]) < 1e-4
    one.close()


def test_replicated_theta0_with_several_ranks_equals_one_rank(m):
    """Option "theta0_shard" = 0: every rank forms the whole theta_0 gradient (through the factor of
    cov, nothing exchanged per evaluation) while the individuals stay sharded. Two and four ranks
    as threads: theta_0 is then the one-rank result to the rounding of the all-reduced precision,
    and bit-identical on all ranks."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_arrays
    a = simu_arrays(203, 30, 450, seed=31)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    th0, thi = np.array([1.0, 1.0]), np.tile([1.0, 1.0, 0.2], (203, 1))

    def job(eng):
        return eng.em_iteration(th0, thi, 0.3, False)
    one = NativeEngine(ds)
    ref = job(one)
    one.close()
    try:
        _lib.call("magma_set_option", b"theta0_shard", 0)
        for world in (2, 4):
            res = _run_ranks(ds, world, job)
            for r in res:
                assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[1], res[0][1]) and r[2:] == res[0][2:]
            assert np.abs(res[0][0] - ref[0]).max() < 1e-7, (world, res[0][0], ref[0])
            assert abs(res[0][2] - ref[2]) < 1e-11 * abs(ref[2]) and abs(res[0][3] - ref[3]) < 1e-6 * abs(ref[3])
    finally:
        _lib.call("magma_set_option", b"theta0_shard", 1)


def test_sharded_theta0_gradient_equals_unsharded(m):
    """One theta_0 evaluation with its products split 1-, 2-, 5-ways agrees to summation order:
    checked through the M-step's first iterate with maxit = 1 (the optimiser's first step is a
    fixed multiple of the gradient at the start)."""
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_arrays
    a = simu_arrays(64, 30, 700, seed=5)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    th0, thi = np.array([1.2, 0.6]), np.tile([1.0, 1.0, 0.3], (64, 1))

    def job(eng):
        assert eng.e_step(th0, thi, 0.3, 0.0) == 0
        return eng.m_step(th0, thi, False, {"maxit": 1})[0]
    one = NativeEngine(ds)
    ref = job(one)
    one.close()
    for world in (2, 5):
        for r in _run_ranks(ds, world, job):
            assert np.abs(r - ref).max() < 1e-11 * max(1.0, np.abs(ref).max()), (world, r, ref)


def test_rccl_communicator_with_one_rank(m):
    """RCCL itself on the one GPU of this box: unique id, ncclCommInitRank (one rank), and every
    all-reduce of an EM iteration going through ncclAllReduce -- packed precision tiles, weighted
    mean, the sharded theta_0 gradient sums (here: all block rows on rank 0), gathered theta_i.
    With one rank the sums are identities, so the result must equal the communicator-free run."""
    from magma_amd.native import NativeEngine, unique_id
    from magma_amd.synth import simu_arrays
    a = simu_arrays(200, 30, 500, seed=3)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    th0, thi = np.array([1.0, 1.0]), np.tile([1.0, 1.0, 0.2], (200, 1))
    uid = unique_id()
    assert len(uid) == 128 and any(uid)
    plain, rccl = NativeEngine(ds), NativeEngine(ds, rank=0, world=1, unique_id=uid)
    for common in (False, True):
        x = plain.em_iteration(th0, thi, 0.2, common)
        y = rccl.em_iteration(th0, thi, 0.2, common)
        assert np.array_equal(plain.get_cov(), rccl.get_cov())
        assert np.array_equal(x[1], y[1]) and x[2] == y[2]
        # theta_0: the sharded route sums the same per-tile pairs in the same order
        assert np.abs(x[0] - y[0]).max() < 1e-9 and abs(x[3] - y[3]) < 1e-10 * abs(x[3])
    plain.close()
    rccl.close()


def test_posterior_mu_grid_is_the_union_over_all_ranks(m):
    """t_pred = sort(unique(db U new_db U timestamps)) is the union over the WHOLE db
    (MG:114). Two and three ranks as threads whose shards hold DISJOINT timestamps, each passing
    only a part of the extra timestamps: every rank must end with the one-rank grid, mean and cov."""
    import ctypes as C
    from magma_amd import _lib
    from magma_amd.native import ALLREDUCE_FN
    rng = np.random.default_rng(11)
    sizes = [9, 14, 7, 12, 10, 8]
    ids, ts, ys = [], [], []
    for k, n in enumerate(sizes):  # individual k on its own residue class of a 0.01 lattice: disjoint
        t = np.sort(rng.choice(np.arange(k, 600, len(sizes)), n, replace=False)) * 0.01
        ids += [k + 1] * n
        ts.append(t)
        ys.append(rng.normal(0, 1, n) + t)
    db = {"ID": np.array(ids), "Timestamp": np.concatenate(ts), "Output": np.concatenate(ys)}
    ds = m.Dataset(db)
    extra = np.array([0.005, 2.505, 5.995, 3.333])
    th0 = np.array([0.8, 0.6])
    thi = np.tile([0.7, 0.5, 0.3], (ds.M, 1)) + rng.uniform(-0.1, 0.1, (ds.M, 3))
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    ref = orc.posterior_mu(db, {"Timestamp": extra[:2]}, extra[2:], 0.3, orc.kernel, orc.kernel, hp)
    hpv = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    m0 = np.array([0.3])

    for world in (2, 3):
        shared = _Shared(world)
        out, errs = [None] * world, []

        def work(r):
            try:
                import torch
                torch.cuda.set_device(0)
                ar = _make_allreduce(shared, r)

                def cb(user, ptr, count, stream):
                    ar(ptr, count)
                    return 0
                keep = ALLREDUCE_FN(cb)
                ctx, dset = C.c_void_p(0), C.c_void_p(0)
                _lib.call("magma_ctx_create_callback", C.byref(ctx), 0, r, world, C.cast(keep, C.c_void_p), None)
                per = (ds.M + world - 1) // world
                lo, hi = min(ds.M, r * per), min(ds.M, r * per + per)
                o0, o1 = int(ds.off[lo]), int(ds.off[hi])
                off = np.ascontiguousarray(ds.off[lo:hi + 1] - o0, dtype=np.int64)
                t, y = np.ascontiguousarray(ds.t[o0:o1]), np.ascontiguousarray(ds.y[o0:o1])
                mine = np.ascontiguousarray(extra[r::world])  # a split of the extra timestamps
                N, info = C.c_int(0), C.c_int(0)
                _lib.call("magma_posterior_mu", ctx, C.byref(dset), hi - lo, ds.M, lo, hpv(off), hpv(t), hpv(y),
                          mine.size, hpv(mine), hpv(th0), hpv(np.ascontiguousarray(thi)), 3, 0.1, hpv(m0),
                          C.byref(N), C.byref(info))
                grid, mean = np.empty(N.value), np.empty(N.value)
                cov = np.empty((N.value, N.value))
                _lib.call("magma_dataset_grid", dset, hpv(grid))
                _lib.call("magma_get_mean", ctx, dset, hpv(mean))
                _lib.call("magma_get_cov", ctx, dset, hpv(cov), N.value)
                out[r] = (info.value, grid, mean, cov)
                _lib.load().magma_dataset_destroy(dset)
                _lib.load().magma_ctx_destroy(ctx)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
                shared.barrier.abort()
        ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join()
        if errs:
            raise errs[0]
        for info, grid, mean, cov in out:
            assert info == 0 and np.array_equal(grid, ref["cov"][0])
            assert np.abs(cov - ref["cov"][1]).max() < 1e-9 * np.abs(ref["cov"][1]).max()
            assert np.abs(mean - ref["mean"]["Output"]).max() < 1e-9 * np.abs(ref["mean"]["Output"]).max()
            assert np.array_equal(cov, out[0][3]) and np.array_equal(mean, out[0][2])


# ------------------------------------------------------------------ prediction side
def test_native_posterior_mu_pred_gp_and_train_new_gp(m):
    import ctypes as C
    from magma_amd import _lib
    from magma_amd.native import train_new_gp
    truth, test, train = dataset("data_FT", 7)
    hp = hp_of("hp_FT", 7)
    ds = m.Dataset(train)
    th0 = hp["theta_0"]
    thi = np.array([hp["theta_i"][int(i)] for i in ds.ids])
    db_obs = {k: v[:20] for k, v in test.items()}
    ts = test["Timestamp"][20:]
    extra = np.concatenate([ts, db_obs["Timestamp"]])
    hpv = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    ctx, dset = C.c_void_p(0), C.c_void_p(0)
    _lib.call("magma_ctx_create", C.byref(ctx), 0)
    N, info = C.c_int(0), C.c_int(0)
    m0 = np.array([0.0])
    _lib.call("magma_posterior_mu", ctx, C.byref(dset), ds.M, ds.M, 0, hpv(ds.off), hpv(ds.t), hpv(ds.y),
              extra.size, hpv(np.ascontiguousarray(extra)), hpv(th0), hpv(np.ascontiguousarray(thi)), 3,
              0.1, hpv(m0), C.byref(N), C.byref(info))
    assert info.value == 0
    grid = np.empty(N.value)
    _lib.call("magma_dataset_grid", dset, hpv(grid))
    mean, cov = np.empty(N.value), np.empty((N.value, N.value))
    _lib.call("magma_get_mean", ctx, dset, hpv(mean))
    _lib.call("magma_get_cov", ctx, dset, hpv(cov), N.value)
    mu = orc.posterior_mu(train, db_obs, ts, 0.0, orc.kernel, orc.kernel, hp)
    assert np.array_equal(grid, mu["cov"][0])
    assert np.abs(cov - mu["cov"][1]).max() < 1e-9 * np.abs(mu["cov"][1]).max()
    assert np.abs(mean - mu["mean"]["Output"]).max() < 1e-9
    # pred_gp against the resident hyper-posterior
    o = np.argsort(db_obs["Timestamp"], kind="stable")
    tn, yn = np.ascontiguousarray(db_obs["Timestamp"][o]), np.ascontiguousarray(db_obs["Output"][o])
    tss = np.ascontiguousarray(np.sort(ts))
    off_n, off_t = np.array([0, tn.size], dtype=np.int64), np.array([0, tss.size], dtype=np.int64)
    th = np.array([1.1, 0.6, 0.3])
    pm, pv, pinfo = np.empty(tss.size), np.empty(tss.size), np.zeros(1, dtype=np.int32)
    _lib.call("magma_pred_gp_resident", ctx, dset, 1, hpv(off_n), hpv(tn), hpv(yn), hpv(off_t), hpv(tss),
              hpv(th), 0, hpv(pm), hpv(pv), hpv(pinfo))
    exp = orc.pred_gp({"Timestamp": tn, "Output": yn}, tss, mu["mean"], mu["cov"], orc.kernel, th[:2], th[2])
    assert pinfo[0] == 0
    assert np.abs(pm - exp["Mean"]).max() < 1e-8 * np.abs(exp["Mean"]).max()
    assert np.abs(pv - exp["Var"]).max() < 1e-8 * np.abs(exp["Var"]).max()
    bad = np.array([0.123])
    rc = _lib.load().magma_pred_gp_resident(ctx, dset, 1, hpv(off_n), hpv(tn), hpv(yn),
                                            hpv(np.array([0, 1], dtype=np.int64)), hpv(bad), hpv(th), 0,
                                            hpv(pm), hpv(pv), hpv(pinfo))
    assert rc == 6 and b"subscript out of bounds" in _lib.load().magma_last_error()
    _lib.load().magma_dataset_destroy(dset)
    _lib.load().magma_ctx_destroy(ctx)
    # train_new_gp natively == the Python-driven route of api.train_new_gp (same optimiser source)
    mean_obs = orc._mean_at(mu["mean"], tn)
    cov_obs = orc._block(mu["cov"][0], mu["cov"][1], tn)
    got, status = train_new_gp(tn, yn, mean_obs, cov_obs, [1.0, 1.0, 0.2])
    ref = m.train_new_gp({"Timestamp": tn, "Output": yn}, mu["mean"], mu["cov"], np.array([1.0, 1.0, 0.2]),
                         m.kernel)
    assert status in (1, 2, 3, 4)
    assert np.abs(got - np.append(ref["theta"], ref["sigma"])).max() < 1e-12


# ------------------------------------------------------------------ n_i > 64
def test_individuals_with_more_than_64_observations(m):
    """kern_to_cov / kern_to_inv accept any length (CF:66-81, 95-104). Individuals with n_i > 64
    take the dense route of the native driver (K_i, its inverse and logL_GP_mod / gr_GP_mod on the
    N x N kernels, scattered with update_inv / update_mean): E-step, M-step objective sums and a
    whole EM iteration against the oracle, in a db that mixes them with small individuals."""
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(65)
    db = _ragged_db(rng, [30, 65, 12, 128, 40, 64, 100])
    ds = m.Dataset(db)
    assert ds.nmax == 128
    th0 = np.array([0.9, 0.7])
    thi = np.tile([1.0, 1.0, 0.4], (ds.M, 1)) + rng.uniform(-0.3, 0.3, size=(ds.M, 3))
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    pen = float(np.mean(thi[:, 2]))
    nat = NativeEngine(ds)
    assert nat.e_step(th0, thi, pen, 0.4) == 0
    exp = orc.e_step(db, 0.4, orc.kernel, orc.kernel, hp)
    sc = np.abs(exp["cov"][1]).max()
    assert np.abs(nat.get_cov() - exp["cov"][1]).max() < 1e-9 * sc
    assert np.abs(nat.get_mean() - exp["mean"]["Output"]).max() < 1e-9 * np.abs(exp["mean"]["Output"]).max()
    ll = orc.logL_monitoring(hp, db, orc.kernel, orc.kernel, exp["mean"], exp["cov"], 0.4)
    assert abs(nat.logL_monitoring(th0, thi) - ll) < 1e-10 * abs(ll)
    # M-step: every individual's optimum against the oracle's m_step driven by the same optimiser
    n0, ni = nat.m_step(th0, thi, False)

    def same_opt(par, fn, gr):
        r = m.minimize_batch(np.asarray(par, float)[None, :],
                             lambda x, act: (np.array([fn(x[0])]), np.array([gr(x[0])])))
        return r["x"][0]
    new = orc.m_step(db, hp, exp["mean"], exp["cov"], orc.kernel, orc.kernel, 0.4, False, optimizer=same_opt)
    ref = np.array([new["theta_i"][i] for i in ds.ids])
    assert np.abs(n0 - new["theta_0"]).max() < 1e-6
    assert np.abs(ni - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), np.abs(ni - ref).max(axis=1)
    # the gradient of the dense route, through the common-HP objective (sum over all individuals)
    a = nat.em_iteration(th0, np.tile([1.0, 1.0, 0.4], (ds.M, 1)), 0.4, True)
    hp_c = {"theta_0": th0, "theta_i": {i: np.array([1.0, 1.0, 0.4]) for i in ds.ids}}
    exp_c = orc.e_step(db, 0.4, orc.kernel, orc.kernel, hp_c)
    new_c = orc.m_step(db, hp_c, exp_c["mean"], exp_c["cov"], orc.kernel, orc.kernel, 0.4, True,
                       optimizer=same_opt)
    assert np.abs(a[1][0] - new_c["theta_i"][ds.ids[0]]).max() < 1e-6
    ll_c = orc.logL_monitoring(hp_c, db, orc.kernel, orc.kernel, exp_c["mean"], exp_c["cov"], 0.4)
    assert abs(a[2] - ll_c) < 1e-9 * abs(ll_c)
    nat.close()
    # the R-named builders on the same db
    got = m.kern_to_inv(db, m.kernel, hp["theta_i"], 0)
    ref_inv = orc.kern_to_inv(db, orc.kernel, hp["theta_i"], 0)
    for i in ds.ids:
        assert np.abs(got[i][1] - ref_inv[i][1]).max() < 1e-9 * np.abs(ref_inv[i][1]).max(), i
    with pytest.raises(ValueError, match="NativeEngine"):
        m.HipBackend(ds, 0, ds.M)


def test_common_times_common_hp_inverts_once(m):
    """BASELINE config 2's shape (all individuals on one grid, common HPs): CF:221-224, 347-350
    invert K once per evaluation; the driver evaluates ONE pseudo-individual with
    C' = C + mean_i(r_i r_i') and multiplies by M (small.hip, dedup_*). Against the batched sum over
    individuals (option dedup = 0) and the oracle's *_common_hp functions."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_db
    db, a = simu_db(M=60, n=30, N=30, common_times=True, common_hp=True, seed=12)
    ds = m.Dataset(db)
    th0 = np.array([0.9, 0.7])
    thi = np.tile([0.8, 0.6, 0.3], (ds.M, 1))
    out = {}
    try:
        for dd in (1, 0):
            _lib.call("magma_set_option", b"dedup", dd)
            eng = NativeEngine(ds)
            assert eng.e_step(th0, thi, 0.3, 0.1) == 0
            eng.reset_stats()
            ll = eng.logL_monitoring(th0, thi)
            n0, ni = eng.m_step(th0, thi, True)
            out[dd] = dict(ll=ll, n0=n0, ni=ni, evals=eng.stats["indiv_evals"],
                           ce=eng.stats["common_evals"], mean=eng.get_mean(), cov=eng.get_cov())
            eng.close()
    finally:
        _lib.call("magma_set_option", b"dedup", 1)
    d1, d0 = out[1], out[0]
    # one evaluation per optimiser step instead of M
    assert d1["ce"] == d0["ce"] and d1["evals"] * ds.M == d0["evals"]
    assert abs(d1["ll"] - d0["ll"]) < 1e-11 * abs(d0["ll"])
    assert np.abs(d1["n0"] - d0["n0"]).max() == 0.0 and np.abs(d1["ni"] - d0["ni"]).max() < 1e-6
    # the objective and gradient themselves against the oracle (CF:203-234, 334-368)
    labels = ds.grid
    mean_db = {"Timestamp": labels, "Output": d1["mean"]}
    hp3 = np.array([0.8, 0.6, 0.3])
    f_o = orc.logL_GP_mod_common_hp(hp3, db, mean_db, orc.kernel, (labels, d1["cov"]))
    hp = {"theta_0": th0, "theta_i": {i: thi[k] for k, i in enumerate(ds.ids)}}
    ll_o = orc.logL_monitoring(hp, db, orc.kernel, orc.kernel, mean_db, (labels, d1["cov"]), 0.1)
    assert abs(d1["ll"] - ll_o) < 1e-9 * abs(ll_o), (d1["ll"], ll_o, f_o)
    # the optimum found is a stationary point of the oracle's summed objective
    g_o = orc.gr_GP_mod_common_hp(d1["ni"][0], db, mean_db, orc.kernel, (labels, d1["cov"]))
    f_new = orc.logL_GP_mod_common_hp(d1["ni"][0], db, mean_db, orc.kernel, (labels, d1["cov"]))
    assert f_new < f_o and np.abs(g_o).max() < 1e-3 * max(1.0, abs(f_new))


def test_failed_estep_factorisation_is_an_error_not_a_silent_nan(m):
    """a K_0 that is not positive definite in floating point (huge length scale, zero
    jitter -- pen_diag is the SIGNED mean of sigma_i, CF:381) must surface: e_step reports
    info != 0 and a NaN f0, the EM drivers stop with LinAlgError (the reference would go on with
    MASS::ginv of that matrix, CF:387 -- not replicated for the N x N factorisations)."""
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(3)
    db = _ragged_db(rng, [20] * 6)
    ds = m.Dataset(db)
    th0 = np.array([1.0, 9.0])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    thi[::2, 2] = -0.2  # mean(sigma_i) = 0, sigma_i^2 > 0
    nat = NativeEngine(ds)
    info = nat.e_step(th0, thi, 0.0, 0.0)
    assert info != 0 and np.isnan(nat.last_f0)
    with pytest.raises(np.linalg.LinAlgError):
        nat.em_iteration(th0, thi, 0.0, False)
    from host_mirror import Engine
    with pytest.raises(np.linalg.LinAlgError):
        Engine(ds).em_iteration(th0, thi, 0.0, False)
    # the engine is still usable afterwards
    assert nat.e_step(np.array([1.0, 1.0]), thi, 0.2, 0.0) == 0 and np.isfinite(nat.last_f0)
    nat.close()


def test_prefetched_estep_sums_are_only_taken_for_the_same_hyperparameters(m):
    """An EM iteration starts the next E-step's sums over individuals under its last monitoring
    call (driver.hip, prefetch_estep_sums). They must be used only when the next E-step is called
    with exactly those theta_i, and give the same bits as an E-step that computes them itself."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(5)
    db = _ragged_db(rng, [30, 12, 40, 7, 25, 33, 1, 64, 50, 50])
    ds = m.Dataset(db)
    th0 = np.array([0.9, 0.7])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1)) + rng.uniform(-0.2, 0.2, size=(ds.M, 3))
    res = {}
    try:
        for pf in (1, 0):
            _lib.call("magma_set_option", b"prefetch", pf)
            eng = NativeEngine(ds)
            n0, ni, _, _ = eng.em_iteration(th0, thi, 0.3, False)
            pen = float(np.mean(ni[:, 2]))
            assert eng.e_step(n0, ni, pen, 0.3) == 0           # the prefetched sums (if any)
            same = (eng.get_mean(), eng.get_cov())
            other = ni + 0.01
            assert eng.e_step(n0, other, pen, 0.3) == 0        # other theta_i: must not be taken
            n1 = eng.em_iteration(n0, ni, 0.3, False)          # and a following iteration
            res[pf] = (n0, ni, same, (eng.get_mean(), eng.get_cov()), n1)
            fresh = NativeEngine(ds)
            assert fresh.e_step(n0, other, pen, 0.3) == 0
            eng.e_step(n0, other, pen, 0.3)
            assert np.array_equal(fresh.get_cov(), eng.get_cov())
            fresh.close()
            eng.close()
    finally:
        _lib.call("magma_set_option", b"prefetch", 1)
    a, b = res[1], res[0]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for x, y in zip(a[2] + a[3], b[2] + b[3]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1]) and a[4][2:] == b[4][2:]


def test_theta0_gradient_through_the_covariance_factor(m):
    """The driver forms K_0^-1 cov K_0^-1 of the theta_0 gradient (CF:304-332) as Y Y' with
    Y = K_0^-1 V', cov = V'V the factor the E-step's inverse leaves behind (N^3 less work than
    through the full product K_0^-1 cov). Same M-step as with the option off, on several tiles."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_arrays
    a = simu_arrays(400, 25, 500, seed=77)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    th0 = np.array([0.8, 0.9])
    thi = np.tile([1.0, 1.0, 0.3], (ds.M, 1))
    out = {}
    try:
        for on in (1, 0):
            _lib.call("magma_set_option", b"cov_factor", on)
            eng = NativeEngine(ds)
            assert eng.e_step(th0, thi, 0.3, 0.0) == 0
            out[on] = eng.m_step(th0, thi, False, {"maxit": 4})
            eng.close()
    finally:
        _lib.call("magma_set_option", b"cov_factor", 1)
    assert np.abs(out[1][0] - out[0][0]).max() < 1e-8, (out[1][0], out[0][0])
    assert np.array_equal(out[1][1], out[0][1])  # (the theta_i problems do not see the option)


def test_speculative_estep_changes_nothing(m):
    """An EM iteration queues the NEXT E-step's sums right behind its resident M-step kernel and that
    E-step's dense part behind its monitoring evaluation (driver.hip: early_estep_sums,
    speculate_estep). With the option off every step is computed when it is asked for: same bits,
    incl. the hyper-posterior left resident after the last iteration (`param`, MG:78), which the
    speculation must not touch. A call with other hyper-parameters than the speculated ones
    ignores it."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    from magma_amd.synth import simu_arrays
    a = simu_arrays(1500, 30, 300, False, False, seed=21)
    ds = m.Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    out = {}
    try:
        for spec in (1, 0):
            _lib.call("magma_set_option", b"speculate", spec)
            eng = NativeEngine(ds)
            th0, thi = np.array([1.0, 1.0]), np.tile([1.0, 1.0, 0.2], (ds.M, 1))
            tr = []
            for _ in range(4):
                th0, thi, lo, ln = eng.em_iteration(th0, thi, 0.3, False)
                tr.append((lo, ln))
            res = [th0.copy(), thi.copy(), np.array(tr), eng.get_mean(), eng.get_cov()]
            # a miss: other theta_0 than the one the last iteration speculated on
            other0 = th0 + np.array([0.05, -0.02])
            assert eng.e_step(other0, thi, float(np.mean(thi[:, 2])), 0.3) == 0
            res += [eng.get_mean(), eng.get_cov(), eng.last_f0]
            # and an iteration from there (the speculation of the iteration before is stale)
            n0, ni, lo, ln = eng.em_iteration(other0, thi, 0.3, False)
            res += [n0, ni, np.array([lo, ln])]
            out[spec] = res
            eng.close()
    finally:
        _lib.call("magma_set_option", b"speculate", 1)
    for x, y in zip(out[1], out[0]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    fresh = NativeEngine(ds)
    th0, thi = out[1][0] + np.array([0.05, -0.02]), out[1][1]
    assert fresh.e_step(th0, thi, float(np.mean(thi[:, 2])), 0.3) == 0
    assert np.array_equal(fresh.get_cov(), out[1][6]) and fresh.last_f0 == out[1][7]
    fresh.close()


@pytest.mark.parametrize("sizes", [[30] * 40, [30, 12, 40, 7, 25, 33, 1, 64, 50, 50, 52, 49]], ids=["uniform", "ragged"])
def test_packed_cov_blocks_gathered_inside_the_resident_kernel(m, sizes):
    """`new_cov[paste0('X', t_i), paste0('X', t_i)]` (CF:188, 312): the driver lets the resident
    M-step kernel gather each individual's block itself (option "resident_gather", default) instead
    of launching magma_indiv_gather_cov between the E-step and the M-step: same bits."""
    from magma_amd import _lib
    from magma_amd.native import NativeEngine
    rng = np.random.default_rng(len(sizes))
    db = _ragged_db(rng, sizes)
    ds = m.Dataset(db)
    out = {}
    try:
        for g in (1, 0):
            _lib.call("magma_set_option", b"resident_gather", g)
            eng = NativeEngine(ds)
            th0 = np.array([0.9, 0.7])
            thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1)) + np.random.default_rng(3).uniform(-0.3, 0.3, size=(ds.M, 3))
            res = []
            for _ in range(3):
                th0, thi, lo, ln = eng.em_iteration(th0, thi, 0.4, False)
                res += [th0.copy(), thi.copy(), np.array([lo, ln])]
            res.append(eng.logL_monitoring(th0, thi))  # (reads the packed blocks after the M-step)
            out[g] = res
            eng.close()
    finally:
        _lib.call("magma_set_option", b"resident_gather", 1)
    for x, y in zip(out[1], out[0]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
