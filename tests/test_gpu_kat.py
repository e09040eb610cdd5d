"""Known-answer tests through the HIP path: the reference's stored results (KAT-1..4) and its
stored trained hyper-parameters (full EM fit) reproduced on the GPU."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from kat_common import run_kat, dataset, hp_of, KATS  # noqa: E402
from oracle import magma_oracle as orc  # noqa: E402  (checker only)

# Relative tolerance on the per-dataset MSE: 1e-8 (north star) for well-conditioned problems.
# The reference inverts with LAPACK LU, the HIP path with Cholesky; two backward-stable
# algorithms agree to ~cond * eps, so datasets whose prior covariance K_0 is ill-conditioned
# (a few stored models have |sigma| ~ 1e-2 and cond(K_0) up to 2e8) get 200 * eps * cond.
TOL = {"KAT1": 1e-8, "KAT2": 1e-8, "KAT3": 1e-7, "KAT4": 1e-8}


def _cond_k0(name, d):
    data_name, hp_name, _, _, pen = KATS[name]
    _, test, train = dataset(data_name, d)
    hp = hp_of(hp_name, d)
    if isinstance(pen, str):
        pen = np.mean([v[2] for v in hp["theta_i"].values()])
    grid = np.unique(np.concatenate([train["Timestamp"], test["Timestamp"]]))
    return np.linalg.cond(orc.kern_to_cov(grid, orc.kernel, hp["theta_0"], pen)[1])


@pytest.fixture(scope="module")
def m():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (no CPU fallback exists)")
    import magma_amd
    return magma_amd


@pytest.mark.parametrize("name", ["KAT1", "KAT2", "KAT3", "KAT4"])
def test_reference_results_reproduced_on_gpu(m, name):
    got, exp = run_kat(name, m.posterior_mu, m.pred_gp, m.kernel)
    rel = np.abs(got[:, 0] - exp[:, 0]) / np.abs(exp[:, 0])
    assert np.median(rel) < 1e-11, (name, np.median(rel))
    loose = np.nonzero(rel >= TOL[name])[0] + 1
    for d in loose:
        lim = 200 * np.finfo(float).eps * _cond_k0(name, int(d))
        assert rel[d - 1] < lim, (name, int(d), rel[d - 1], lim)
    # how many datasets needed the cond(K_0)-scaled bound, and the worst error on each side of it
    # (run with -s / -rP to see it; also kept in the JUnit/"passed" section of the report)
    tight = rel[rel < TOL[name]]
    print(f"{name}: {tight.size}/100 datasets within {TOL[name]:g} (worst {tight.max():.2e}); "
          f"{loose.size} took the 200*eps*cond(K_0) bound"
          + (f" (datasets {list(map(int, loose))}, worst rel err {rel[loose - 1].max():.2e})"
             if loose.size else ""))
    # at most a handful of stored models are that ill-conditioned
    assert loose.size <= 5, (name, list(map(int, loose)))
    # CI95 coverage is a count: identical unless a point sits within rounding of a band edge
    assert np.abs(got[:, 1] - exp[:, 1]).max() < 1e-9, name


@pytest.mark.parametrize("d", [1, 2, 4])
def test_training_common_hp_reproduces_stored_model(m, d):
    """`training(db, 0, ini_hp, kernel_mu, kernel, common_hp = T)` on fixture dataset d of the
    FT setting against the hyper-parameters R stored in train_FT.rds (north star: 1e-8)."""
    _, _, train = dataset("data_FT", d)
    trace = []
    res = m.training(train, 0.0, {"theta_0": [1.0, 1.0], "theta_i": [1.0, 1.0, 0.2]},
                     m.kernel_mu, m.kernel, common_hp=True, trace=trace)
    ref = hp_of("hp_FT", d)
    assert res["convergence"]
    assert np.abs(res["hp"]["theta_0"] - ref["theta_0"]).max() < 1e-7, (res["hp"]["theta_0"], ref["theta_0"])
    for i, th in ref["theta_i"].items():
        assert np.abs(res["hp"]["theta_i"][i] - th).max() < 1e-7


def test_training_individual_hp_close_to_stored_model(m):
    """FF setting (individual HPs): 20 independent L-BFGS-B runs per EM iteration stop at
    factr = 1e7, so agreement with the stored R result is at optimiser tolerance (the CPU
    oracle with scipy's L-BFGS-B agrees with R to ~3e-5 on this dataset)."""
    _, _, train = dataset("data_FF", 1)
    res = m.training(train, 0.0, {"theta_0": [1.0, 1.0], "theta_i": [1.0, 1.0, 0.2]},
                     m.kernel_mu, m.kernel, common_hp=False)
    ref = hp_of("hp_FF", 1)
    assert np.abs(res["hp"]["theta_0"] - ref["theta_0"]).max() < 1e-3
    err = max(np.abs(res["hp"]["theta_i"][i] - th).max() for i, th in ref["theta_i"].items())
    assert err < 5e-3, err
