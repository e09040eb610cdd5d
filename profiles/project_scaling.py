"""Projected strong scaling of the EM iteration over k = 1, 2, 4, 8 GPUs, MEASURED ON ONE GPU
(projection, not a multi-GPU measurement: the one-GPU box cannot run RCCL with more than one rank).

For every k one rank's share of every phase runs alone on the GPU through the library's real
multi-rank code path (callback communicator, world = k, rank 0): M / k individuals for the E-step
sums, the resident M-step and the monitoring sums; the theta_0 chain and the dense inverses
replicated, as on every rank of a real job with option "theta0_shard" = 0 (with the default,
theta0_shard = 1, a rank forms 1 / k of the block rows of the two N^3 products of the theta_0
gradient instead: that needs the other ranks' true partial sums, which one GPU cannot supply, so
the table states the replicated variant and prices the split separately from the products'
measured kernel times). The all-reduce callback stands in for the other k - 1 ranks without moving
data: partial sums over individuals are multiplied by k (the other ranks' shards are statistically
alike), the gathered theta_i rows of the other ranks are filled with copies of rank 0's. What the wire would add is modelled on top:
ring all-reduce of the packed precision tiles + w over xGMI (2 (k - 1) / k x bytes / link
bandwidth) and one latency per small message.

    python profiles/project_scaling.py [--steps 8] [--warmup 2] [--link-gbs 64] [--latency-us 25]

Prints one JSON line per (config, k) and a markdown table for DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--link-gbs", type=float, default=64.0,
                    help="effective xGMI bandwidth per ring direction (GB/s); 7 links x ~153 GB/s "
                         "bidirectional per GPU, a ring uses one link pair")
    ap.add_argument("--latency-us", type=float, default=25.0, help="per small all-reduce")
    ap.add_argument("--configs", default="c3,c5")
    ap.add_argument("--ks", default="1,2,4,8")
    args = ap.parse_args()
    import torch
    from magma_amd.engine import Dataset
    from magma_amd.native import NativeEngine, device_tensor
    from magma_amd.synth import simu_arrays
    from magma_amd import _lib
    import ctypes as C
    lib = _lib.load()
    _lib.call("magma_set_option", b"theta0_shard", 0)
    cfgs = {"c3": dict(M=10000, n=50, N=2000, seed=1003), "c5": dict(M=100000, n=40, N=2000, seed=1005)}
    rows = []
    for name in args.configs.split(","):
        cfg = cfgs[name]
        a = simu_arrays(cfg["M"], cfg["n"], cfg["N"], False, False, seed=cfg["seed"])
        ds = Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
        Mt = ds.M
        for k in [int(x) for x in args.ks.split(",")]:
            msgs = {"big_bytes": 0, "small": 0, "host_s": 0.0}

            def fake(ptr, count, k=k):
                # the driver has synchronised the stream before calling: everything from here to
                # the return is the stand-in's own cost (Python + two tiny kernels + a sync), which
                # an all-reduce enqueued on the stream would not have; it is subtracted below
                t_in = time.perf_counter()
                t = device_tensor(ptr, count)
                if count == 3 * Mt:  # gathered theta_i: the other ranks' rows
                    per = (Mt + k - 1) // k
                    v = t.view(Mt, 3)
                    rest = Mt - per
                    if rest > 0:
                        v[per:] = v[:per].repeat(k, 1)[:rest]
                    msgs["big_bytes"] += 8 * count
                else:
                    t.mul_(float(k))
                    if count > 1024:
                        msgs["big_bytes"] += 8 * count
                    else:
                        msgs["small"] += 1
                torch.cuda.synchronize()
                msgs["host_s"] += time.perf_counter() - t_in
            eng = NativeEngine(ds) if k == 1 else NativeEngine(ds, rank=0, world=k, allreduce=fake)
            ini0, ini_i = np.array([1.0, 1.0]), np.tile([1.0, 1.0, 0.2], (Mt, 1))
            th0, thi = ini0.copy(), ini_i.copy()
            for _ in range(args.warmup):
                th0, thi, _, _ = eng.em_iteration(th0, thi, 0.0, False)
            th0, thi = ini0.copy(), ini_i.copy()
            msgs["big_bytes"] = msgs["small"] = 0
            msgs["host_s"] = 0.0
            eng.reset_stats()
            lib.magma_prof_reset()
            lib.magma_prof_enable(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                th0, thi, _, _ = eng.em_iteration(th0, thi, 0.0, False)
            torch.cuda.synchronize()
            raw_ms = 1e3 * (time.perf_counter() - t0) / args.steps
            standin_ms = 1e3 * msgs["host_s"] / args.steps
            ms = raw_ms - standin_ms
            lib.magma_prof_enable(0)
            prod_ms = 0.0  # the two gradient products of the theta_0 evaluations, per step
            for pid in (0, 1):
                cnt, tot = C.c_int(0), C.c_double(0.0)
                lib.magma_prof_get(pid, C.byref(cnt), C.byref(tot))
                prod_ms += tot.value / args.steps
            st = eng.stats
            eng.close()
            big = msgs["big_bytes"] / args.steps
            small = msgs["small"] / args.steps
            wire = 0.0 if k == 1 else (2.0 * (k - 1) / k * big / (args.link_gbs * 1e9) * 1e3
                                       + small * args.latency_us * 1e-3)
            units = Mt / 10000.0
            row = {"config": name, "k": k, "M_per_gpu": (Mt + k - 1) // k, "compute_ms_per_step": ms,
                   "standin_callback_ms_per_step_subtracted": standin_ms,
                   "allreduce_bytes_per_step": big, "small_allreduces_per_step": small,
                   "modelled_wire_ms_per_step": wire, "projected_ms_per_step": ms + wire,
                   "projected_value_10k_units_per_s": units / ((ms + wire) * 1e-3),
                   "theta0_evals_per_step": st["theta0_evals"] / args.steps,
                   "theta0_products_ms_per_step": prod_ms,
                   # theta0_shard = 1: 3 N^3 / k per rank instead of the 2 N^3 of the factor route,
                   # + one 16-byte all-reduce per evaluation (upper bound of the gain: the products
                   # overlap with the resident M-step in part)
                   "sharded_theta0_ms_per_step_estimate": ms + wire - prod_ms * max(0.0, 1.0 - 1.5 / k)
                   + (0.0 if k == 1 else st["theta0_evals"] / args.steps * args.latency_us * 1e-3),
                   "indiv_evals_per_step_this_rank": st["indiv_evals"] / args.steps,
                   "kind": "projection, unmeasured on k GPUs"}
            rows.append(row)
            print(json.dumps(row), flush=True)
    print()
    print("| config | k | individuals / GPU | one rank's share alone (ms / EM it) | modelled xGMI (ms) | projected ms / EM it | projected EM it/s (10k-task units) | speed-up vs k = 1 | with the theta_0 products split (estimate, ms) |")
    print("|---|---|---|---|---|---|---|---|---|")
    base = {}
    for r in rows:
        base.setdefault(r["config"], r["projected_ms_per_step"])
        print(f"| {r['config']} | {r['k']} | {r['M_per_gpu']} | {r['compute_ms_per_step']:.2f} | "
              f"{r['modelled_wire_ms_per_step']:.2f} | {r['projected_ms_per_step']:.2f} | "
              f"{r['projected_value_10k_units_per_s']:.1f} | {base[r['config']] / r['projected_ms_per_step']:.2f}x | "
              f"{r['sharded_theta0_ms_per_step_estimate']:.2f} |")


if __name__ == "__main__":
    main()
