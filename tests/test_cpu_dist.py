"""The sharding + reduction PATTERN of the multi-rank EM under gloo, world_size 2, on
the CPU: contiguous shards of individuals, all-reduced partial precision / weighted mean, summed
objective pieces, gathered per-individual hyper-parameters.

What this covers and what it does not: no GPU here, so the math is the oracle's (tests/cpu_backend)
and the EM logic is the Python MIRROR of the driver (tests/host_mirror.py) -- test infrastructure,
not the product. The product's multi-rank path (csrc/driver.hip behind NativeEngine) is covered on
the GPU: two gloo PROCESSES sharing one GPU through the library's callback communicator
(tests/test_gpu_dist_native.py), several ranks as threads (tests/test_gpu_native.py), and RCCL with
one rank; more than one RCCL rank needs more than one GPU and is the driver's SCALE run."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _problem():
    from magma_amd.synth import simu_arrays
    a = simu_arrays(M=7, n=8, N=40, seed=5)
    return a


def _run(rank, world, port, common, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from magma_amd.engine import Dataset
    from host_mirror import Engine, Comm
    from cpu_backend import OracleBackend
    comm = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = Comm(dist)
    a = _problem()
    ds = Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    eng = Engine(ds, backend_factory=lambda d, lo, hi: OracleBackend(d, lo, hi), comm=comm)
    th0 = np.array([1.0, 1.0])
    thi = np.tile([1.0, 1.0, 0.2], (ds.M, 1))
    res = []
    for _ in range(2):
        th0, thi, l_old, l_new = eng.em_iteration(th0, thi, 0.5, common, {"maxit": 6})
        res.append((l_old, l_new))
    if rank == 0:
        np.savez(out, th0=th0, thi=thi, ll=np.array(res), mean=eng.be.get_mean())
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("common", [False, True])
def test_two_ranks_match_one_rank(tmp_path, common):
    f1 = str(tmp_path / "w1.npz")
    f2 = str(tmp_path / "w2.npz")
    _run(0, 1, 0, common, f1)
    port = 29500 + (os.getpid() % 2000) + (1 if common else 0)
    mp.spawn(_run, args=(2, port, common, f2), nprocs=2, join=True)
    a, b = np.load(f1), np.load(f2)
    assert np.abs(a["mean"] - b["mean"]).max() < 1e-10
    assert np.abs(a["ll"] - b["ll"]).max() < 1e-8 * np.abs(a["ll"]).max()
    assert np.abs(a["th0"] - b["th0"]).max() < 1e-8
    assert np.abs(a["thi"] - b["thi"]).max() < 1e-7
