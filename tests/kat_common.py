"""Shared harness for the known-answer tests (KAT-1..4 of SURVEY.md §4.2).

`run_kat(name, posterior_mu, pred_gp, datasets)` re-enacts the reference's evaluation loops
(`eval_mu`, simulation_study.R:396-431, and `loop_pred`, :275-324, with nb_obs=20, nb_test=10)
on the fixture datasets through the given implementations and returns (got, expected) arrays
of (MSE, Ratio_IC) per dataset. The same harness drives the CPU oracle and the HIP path.
"""
import os
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (data file, hp file, expected key, kind, pen_diag used when the results were made)
KATS = {
    "KAT1": ("data_FF", "hp_FF", "res_mu_FF", "mu", "mean_sigma"),
    "KAT2": ("data_FT", "hp_FT", "res_mu_FT", "mu", "mean_sigma"),
    "KAT3": ("data_FT", "hp_FT", "res_pred_FT", "pred", "mean_sigma"),
    "KAT4": ("data_FF", "hp_FTonFFdata", "res_pred_FTonFFdata", "pred", 0.1),
}
_cache = {}


def _load(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLD, name + ".npz")))
    return _cache[name]


def dataset(data_name, d):
    """Rows of fixture dataset d (1-based) split into truth (ID 0), test (ID 1), train."""
    z = _load(data_name)
    sel = z["ID_dataset"] == d
    ids, t, y = z["ID"][sel].astype(np.int64), z["Timestamp"][sel], z["Output"][sel]

    def sub(mask):
        return {"ID": ids[mask], "Timestamp": t[mask], "Output": y[mask]}
    return sub(ids == 0), sub(ids == 1), sub(ids > 1)


def hp_of(hp_name, d):
    z = _load(hp_name)
    return {"theta_0": z["theta_0"][d - 1].copy(),
            "theta_i": {int(i): z["theta_i"][d - 1, k].copy() for k, i in enumerate(z["ids"])}}


def eval_algo(pred_mean, pred_var, db_test):
    """Algo row of eval_methods (simulation_study.R:188-204) with R's recycling."""
    pred_mean = np.asarray(pred_mean, float)
    sd = np.sqrt(np.asarray(pred_var, float))
    obs = np.resize(np.asarray(db_test, float), pred_mean.size)
    mse = np.mean(np.abs(pred_mean - obs) ** 2)
    nb = np.sum((pred_mean - 1.96 * sd < obs) & (obs < pred_mean + 1.96 * sd))
    return mse, nb / np.size(db_test) * 100.0


def run_kat(name, posterior_mu, pred_gp, kernel, datasets=range(1, 101)):
    data_name, hp_name, exp_key, kind, pen = KATS[name]
    expected = _load("expected_results")[exp_key]
    got = []
    for d in datasets:
        truth, test, train = dataset(data_name, d)
        hp = hp_of(hp_name, d)
        if kind == "mu":
            t_pred = train["Timestamp"][np.sort(np.unique(train["Timestamp"], return_index=True)[1])]
            res = posterior_mu(train, test, t_pred, 0.0, kernel, kernel, hp, pen_diag=pen)
            truth_at = truth["Output"][np.isin(truth["Timestamp"], t_pred)]
            got.append(eval_algo(res["pred_GP"]["Mean"], res["pred_GP"]["Var"], truth_at))
        else:
            order = np.argsort(test["Timestamp"], kind="stable")
            obs_sel = np.sort(order[:20])      # top_n(-20, Timestamp), original row order
            pred_sel = np.sort(order[-10:])    # top_n(10, Timestamp)
            db_obs = {k: v[obs_sel] for k, v in test.items()}
            db_pred = {k: v[pred_sel] for k, v in test.items()}
            t_i_pred = db_pred["Timestamp"]
            mu = posterior_mu(train, db_obs, t_i_pred, 0.0, kernel, kernel, hp, pen_diag=pen)
            first = next(iter(hp["theta_i"].values()))
            pred = pred_gp(db_obs, t_i_pred, mu["mean"], mu["cov"], kernel, first[:2], first[2])
            got.append(eval_algo(pred["Mean"], pred["Var"], db_pred["Output"]))
    got = np.array(got)
    exp = expected[np.array(list(datasets)) - 1]
    return got, exp
