"""The product's multi-rank path under torch.distributed: two gloo PROCESSES share the one GPU, each
a rank of the native driver (magma_amd.native.engine_for_job: the library's callback communicator
over the process group; RCCL needs one GPU per rank, which the test box does not have). The
R-named API (`magma_amd.training`, `e_step`) is called the same way on every rank, as a
torch.distributed user would. Against the same calls in one process."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _problem():
    from magma_amd.synth import simu_db
    db, _ = simu_db(M=11, n=12, N=60, common_times=False, common_hp=False, seed=8)
    return db


def _run(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import magma_amd as m
    from magma_amd.native import engine_for_job
    torch.cuda.set_device(0)
    db = _problem()
    ds = m.Dataset(db)
    eng = engine_for_job(ds)
    assert eng.world == world and eng.rank == rank
    eng.selfcheck(20000)
    th0 = np.array([1.0, 1.0])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    res = []
    for _ in range(2):
        th0, thi, l_old, l_new = eng.em_iteration(th0, thi, 0.5, False)
        res.append((l_old, l_new))
    mean = eng.get_mean()
    s0, g0 = eng.sum_fi(np.array([1.0, 0.8, 0.3]), True)  # CF:203-234, 334-368
    eng.close()
    # the R-named API, called alike on every rank
    ini = {"theta_0": np.array([1.0, 1.0]), "theta_i": np.array([1.0, 1.0, 0.2])}
    fit = m.training(db, 0.5, ini, m.kernel_mu, m.kernel, common_hp=True)
    par = m.e_step(db, 0.5, m.kernel_mu, m.kernel, fit["hp"])
    if rank == 0:
        np.savez(out, th0=th0, thi=thi, ll=np.array(res), mean=mean, s0=s0, g0=g0,
                 fit0=fit["hp"]["theta_0"], fiti=np.array([fit["hp"]["theta_i"][i] for i in ds.ids]),
                 pmean=par["mean"]["Output"], pcov=par["cov"][1])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_gloo_processes_on_one_gpu_match_one_process(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import torch.multiprocessing as mp
    f1, f2 = str(tmp_path / "w1.npz"), str(tmp_path / "w2.npz")
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_run, args=(2, port, f2), nprocs=2, join=True)
    mp.spawn(_run, args=(1, 0, f1), nprocs=1, join=True)
    a, b = np.load(f1), np.load(f2)
    # sums over individuals are formed shard by shard and added: last-bit differences, which the
    # optimisers amplify to their own tolerance (see tests/test_gpu_trajectory.py)
    assert np.abs(a["mean"] - b["mean"]).max() < 1e-6 * max(1.0, np.abs(a["mean"]).max())
    assert abs(a["ll"][0, 0] - b["ll"][0, 0]) < 1e-10 * abs(a["ll"][0, 0])
    assert np.abs(a["ll"] - b["ll"]).max() < 1e-6 * np.abs(a["ll"]).max()
    assert np.abs(a["th0"] - b["th0"]).max() < 1e-5 and np.abs(a["thi"] - b["thi"]).max() < 1e-4
    assert abs(a["s0"] - b["s0"]) < 1e-6 * abs(a["s0"])
    assert np.abs(a["g0"] - b["g0"]).max() < 1e-6 * np.abs(a["g0"]).max()
    assert np.abs(a["fit0"] - b["fit0"]).max() < 1e-5 and np.abs(a["fiti"] - b["fiti"]).max() < 1e-4
    assert np.abs(a["pmean"] - b["pmean"]).max() < 1e-5 * max(1.0, np.abs(a["pmean"]).max())
    assert np.abs(a["pcov"] - b["pcov"]).max() < 1e-5 * np.abs(a["pcov"]).max()
