"""Extract known-answer fixtures from the reference's stored simulation study.

Run once in the build container (where /root/reference exists); writes tests/golden/*.npz.
Only DATA is extracted (datasets, trained hyper-parameters, per-dataset result rows) — no
reference source. Inputs:
  Simulations/Data/db_rep_100_M_20_N_30_time_FALSE_hp_{FALSE,TRUE}.csv  (';' sep, ',' decimal)
  Simulations/Training/train_{FF,FT,FTonFFdata}.rds                    (decoded by rds_reader)
  Simulations/Results/res_{mu,pred}_N20-10_M20_*.csv                   (Algo rows)
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
from rds_reader import read_rds_toplevel_list  # noqa: E402

REF = "/root/reference/Simulations"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def num(s):
    s = s.strip().strip('"')
    return float("nan") if s == "NA" else float(s.replace(",", "."))


def read_data(path):
    ids, ts, ys, ds = [], [], [], []
    with open(path) as f:
        header = f.readline().strip().split(";")
        assert header[:3] == ["ID", "Timestamp", "Output"] and header[-1] == "ID_dataset"
        for line in f:
            p = line.strip().split(";")
            ids.append(int(p[0]))
            ts.append(num(p[1]))
            ys.append(num(p[2]))
            ds.append(int(p[-1]))
    return (np.array(ids, np.int16), np.array(ts), np.array(ys), np.array(ds, np.int16))


def read_hp(path):
    meta, items = read_rds_toplevel_list(path)
    assert meta["writer"] == "0x30603"
    th0 = np.zeros((100, 2))
    thi = np.zeros((100, 20, 3))
    ttrain = np.zeros(100)
    ids = None
    for d in range(100):
        algo = items[d].as_dict()["algo"].as_dict()
        hp = algo["hp"].as_dict()
        th0[d] = [c.value[0] for c in hp["theta_0"].value]
        names = hp["theta_i"].names()
        if ids is None:
            ids = [int(x) for x in names]
        assert [int(x) for x in names] == ids
        for k, df in enumerate(hp["theta_i"].value):
            thi[d, k] = [c.value[0] for c in df.value]
        ttrain[d] = algo["Time_train"].value[0]
    ini = items[101].as_dict()
    assert items[100].value == [0.0]  # prior_mean
    assert ini["theta_0"].value == [1.0, 1.0] and ini["theta_i"].value == [1.0, 1.0, 0.2]
    return th0, thi, np.array(ids, np.int16), ttrain


def read_algo_rows(path):
    rows = []
    with open(path) as f:
        f.readline()
        for line in f:
            p = line.strip().split(";")
            if p[4].strip('"') == "Algo":
                rows.append([num(p[0]), num(p[1])])
    return np.array(rows)


def main():
    os.makedirs(OUT, exist_ok=True)
    for tag, fn in (("FF", "db_rep_100_M_20_N_30_time_FALSE_hp_FALSE.csv"),
                    ("FT", "db_rep_100_M_20_N_30_time_FALSE_hp_TRUE.csv")):
        ids, ts, ys, ds = read_data(f"{REF}/Data/{fn}")
        np.savez_compressed(f"{OUT}/data_{tag}.npz", ID=ids, Timestamp=ts, Output=ys, ID_dataset=ds)
        print(tag, ids.shape)
    for tag in ("FF", "FT", "FTonFFdata"):
        th0, thi, ids, ttrain = read_hp(f"{REF}/Training/train_{tag}.rds")
        np.savez_compressed(f"{OUT}/hp_{tag}.npz", theta_0=th0, theta_i=thi, ids=ids,
                            Time_train=ttrain)
        print(tag, th0[0], thi[0, 0], ttrain[0])
    exp = {
        "res_mu_FF": read_algo_rows(f"{REF}/Results/res_mu_N20-10_M20_FF.csv"),
        "res_mu_FT": read_algo_rows(f"{REF}/Results/res_mu_N20-10_M20_FT.csv"),
        "res_pred_FT": read_algo_rows(f"{REF}/Results/res_pred_N20-10_M20_FT.csv"),
        "res_pred_FTonFFdata": read_algo_rows(f"{REF}/Results/res_pred_N20-10_M20_FTonFFdata.csv"),
    }
    for k, v in exp.items():
        assert v.shape == (100, 2), (k, v.shape)
    np.savez_compressed(f"{OUT}/expected_results.npz", **exp)
    print({k: v[0] for k, v in exp.items()})


if __name__ == "__main__":
    main()
