"""bench.py -- EM iterations/s (E+M) of the MAGMA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c4|c5|tiny]

A "step" is one pass of the body of `training`'s loop (MAGMA.R:32-75): E-step, M-step (L-BFGS-B
on theta_0 and on every individual's theta_i), and the two logL_monitoring calls -- continuing
the EM trajectory from the reference's initial hyper-parameters on synthetic data of the
reference's simulation recipe. Default workload = BASELINE.json configs[2] ("c3"): M = 10,000
individuals x 50 points per GPU, union grid N = 2,000, individual hyper-parameters.

The K timed iterations start again from the reference's initial hyper-parameters after the W
warmup iterations (--trajectory restart, the default), so the expensive first iterations of a
fit are inside the timed region; `fit` reports the iterations up to the reference's stopping rule
(eps < 1e-3) on their own, `steady_state` the cheap tail; --fit times one training() call.

With N GPUs (launched by torch.distributed.run, one rank per GPU; the EM's collectives run on the
library's own RCCL communicator) every rank holds 10,000 individuals (--scaling weak, default), or
the N ranks share the configuration's M (--scaling strong: c3 = the north star's M = 10^4 fit,
c5 = BASELINE's M = 10^5); `value` is the whole-job rate in units of "EM iterations of a 10k-task
problem per second" = steps * (M_total / 10,000) / seconds, so that N = 1 is exactly BASELINE.json's
metric.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle = NumPy/LAPACK restatement of the R code,
timed on a bounded sample on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {  # per-GPU M, n, N, common_times, common_hp, seed
    "tiny": dict(M=200, n=30, N=200, common_times=False, common_hp=False, seed=1001),
    "c2": dict(M=1000, n=50, N=50, common_times=True, common_hp=True, seed=1002),
    "c3": dict(M=10000, n=50, N=2000, common_times=False, common_hp=False, seed=1003),
    "c4": dict(M=2000, n=50, N=8000, common_times=False, common_hp=False, seed=1004),
    "c5": dict(M=12500, n=40, N=2000, common_times=False, common_hp=False, seed=1005),
}
PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X vendor figure; dense, no sparsity
PEAK_HBM_GBS = 8000.0
PROF_NAMES = ["gemm_nt(T=Kinv*C)", "theta0_contract", "spd_inverse", "indiv_eval(f+grad)",
              "indiv_eval(f)", "estep_scatter", "se_cov_grid", "se_cov_batched", "se_inv_batched",
              "indiv_gather_cov", "indiv_mstep(resident)"]


def pmc_traffic(kernel_class, args):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    THIS command line (profiles/r03_pmc_<config>_<steps>_<warmup>.json, made by
    profiles/profile_round.sh: FETCH_SIZE and WRITE_SIZE collected in separate runs; counters in
    KB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced loads on gfx950).
    Counters cannot be read inside a timed run; a file made with other --steps / --warmup /
    --config is not used: then the figure is null."""
    prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    name = next((f"{tag}_pmc_{args.config}_{args.steps}_{args.warmup}.json" for tag in ("r03", "r02")
                 if os.path.exists(os.path.join(prof_dir, f"{tag}_pmc_{args.config}_{args.steps}_{args.warmup}.json"))),
                None)
    keys = {"indiv_mstep(resident)": "magma::indiv_mstep_kernel<",
            "spd_inverse": "magma::spd_inverse_dag_kernel",
            "gemm_nt(T=Kinv*C)": "magma::gemm_nt_kernel"}
    if kernel_class not in keys or name is None:
        return None, None
    path = os.path.join(prof_dir, name)
    with open(path) as fh:
        pmc = json.load(fh)
    key = next((k for k in pmc.get("FETCH_SIZE", {}) if k.startswith(keys[kernel_class])), None)
    try:
        kb = 2.0 * pmc["FETCH_SIZE"][key]["avg_kb"] + pmc["WRITE_SIZE"][key]["avg_kb"]
    except KeyError:
        return None, None
    return kb * 1024.0, "profiles/" + name


def cpu_baseline(cfg):
    """Oracle (NumPy/LAPACK port of the reference's operation sequence) on a bounded sample of
    the same workload, about 10-20 s of CPU work: the dense parts at the full grid size N
    (repeated for a stable time), the per-individual parts on `Ms` individuals; the caller
    combines them at the evaluation counts of the GPU run (per-individual loops are
    independent in the reference, so they scale linearly to M)."""
    from oracle import magma_oracle as orc  # baseline leg only
    from magma_amd.synth import simu_arrays
    N, n = cfg["N"], cfg["n"]
    Ms = 200
    a = simu_arrays(Ms, n, N, cfg["common_times"], cfg["common_hp"], seed=cfg["seed"])
    ids = np.repeat(np.arange(Ms), n)
    db = {"ID": ids, "Timestamp": a["t"], "Output": a["y"]}
    hp = {"theta_0": np.array([1.0, 1.0]), "theta_i": {i: np.array([1.0, 1.0, 0.2]) for i in range(Ms)}}
    reps_dense = max(1, min(20, int(20 * (2000.0 / N) ** 3)))
    t0 = time.time()
    for _ in range(reps_dense):
        par = orc._hyperposterior(db, a["grid"], 0.0, orc.kernel, orc.kernel, hp, 0.2)
    t_e = (time.time() - t0) / reps_dense
    labels, cov = par["cov"]
    # theta_0 objective + gradient (L-BFGS-B calls both at every point)
    t0 = time.time()
    for _ in range(2 * reps_dense):
        orc.logL_GP_mod(hp["theta_0"], par["mean"], 0.0, orc.kernel, cov, 0.4)
        orc.gr_GP_mod(hp["theta_0"], par["mean"], 0.0, orc.kernel, cov, 0.4)
    t_0 = (time.time() - t0) / (2 * reps_dense)
    # per-individual objective + gradient, incl. the reference's per-call block extraction
    blocks = []
    for i in range(Ms):
        sel = ids == i
        db_i = {"Timestamp": a["t"][sel], "Output": a["y"][sel]}
        blocks.append((db_i, orc._mean_at(par["mean"], db_i["Timestamp"]),
                       orc._block(labels, cov, db_i["Timestamp"])))
    t0 = time.time()
    for _ in range(20):
        for i, (db_i, m_i, c_i) in enumerate(blocks):
            orc.logL_GP_mod(hp["theta_i"][i], db_i, m_i, orc.kernel, c_i)
            orc.gr_GP_mod(hp["theta_i"][i], db_i, m_i, orc.kernel, c_i)
    t_i = (time.time() - t0) / (20 * Ms)
    t0 = time.time()
    for _ in range(5):
        orc.kern_to_inv(db, orc.kernel, hp["theta_i"], 0)
    t_ei = (time.time() - t0) / (5 * Ms)
    return {"t_estep_dense": max(t_e - t_ei * Ms, 0.0), "t_estep_per_indiv": t_ei,
            "t_theta0_eval": t_0, "t_indiv_eval": t_i, "Ms": Ms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N GPUs: weak = the config's M per GPU; strong = the config's M in total "
                         "(c3: the north star's M = 10^4 fit; c5: M = 10^5), sharded over the ranks")
    ap.add_argument("--trajectory", default="restart", choices=["restart", "continue"],
                    help="restart (default): the K timed EM iterations start again from the "
                         "reference's initial hyper-parameters after the warmup iterations (so the "
                         "expensive first iterations of a fit are inside the timed region); "
                         "continue: they go on where the warmup stopped (steady state)")
    ap.add_argument("--fit", action="store_true",
                    help="time ONE call of training() (MAGMA.R:15-82) to the reference's stopping "
                         "rule instead of K iterations")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (MAGMA_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks --
    # every rank on GPU local % device_count, the launcher's collectives on gloo / CPU tensors, the
    # EM's all-reduces through the library's callback communicator staged through the host)
    backend = os.environ.get("MAGMA_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    ctl = "cuda" if backend == "nccl" else "cpu"  # where the launcher-level control tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from magma_amd import _lib
    from magma_amd.engine import Dataset
    from magma_amd.native import NativeEngine, unique_id
    from magma_amd.synth import simu_arrays
    lib = _lib.load()
    if os.environ.get("MAGMA_DEDUP", "1") == "0":  # C2: batched sum instead of one inverse per evaluation
        _lib.call("magma_set_option", b"dedup", 0)
    for kv in filter(None, os.environ.get("MAGMA_OPTIONS", "").split(",")):  # e.g. "dag_walker=0,prefetch=0"
        name, val = kv.split("=")
        _lib.call("magma_set_option", name.strip().encode(), int(val))

    cfg = dict(CONFIGS[args.config])
    if args.config == "c5" and args.scaling == "strong":
        cfg["M"] = 100000  # BASELINE config 5: M = 100,000 in total, sharded 1/2/4/8 ways
    M_total = cfg["M"] * (world if args.scaling == "weak" else 1)
    a = simu_arrays(M_total, cfg["n"], cfg["N"], cfg["common_times"], cfg["common_hp"],
                    seed=cfg["seed"])
    ds = Dataset.from_arrays(a["off"], a["t"], a["y"], grid=a["grid"])
    comm_kind = "none" if world == 1 else "rccl (inside libmagma_hip.so)"
    dev = torch.device("cuda", local)

    def make_engine(dataset, note=None):
        """This rank's engine on `dataset` with the job's communicator: the library's own RCCL
        communicator, checked with a known vector on both of the library's streams before anything
        is timed (magma_ctx_selfcheck: polled against a deadline, never a blocking wait); if RCCL
        cannot be created inside the library, or sums wrongly, the library's callback communicator
        on the launcher's process group takes over (same EM code path; reported in
        config.communicator). A collective that does not complete leaves a blocked stream behind:
        reported, exit code 3 -- a process that has touched the GPU is never replaced by another."""
        nonlocal comm_kind
        if world == 1:
            return NativeEngine(dataset, rank=rank, world=world, device=dev)
        e = None
        if (os.environ.get("MAGMA_BENCH_COMM", "rccl") == "rccl" and backend == "nccl"
                and not comm_kind.startswith("callback")):
            box = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                try:
                    box.copy_(torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8))
                except _lib.MagmaHipError as err:
                    print(f"[bench] ncclGetUniqueId inside the library failed: {err}", file=sys.stderr)
            dist.broadcast(box, 0)
            uid = bytes(box.cpu().numpy().tobytes())
            status = 1
            if any(uid):
                try:
                    e = NativeEngine(dataset, rank=rank, world=world, unique_id=uid, device=dev)
                    e.selfcheck(int(os.environ.get("MAGMA_BENCH_COMM_TIMEOUT_MS", "30000")))
                    status = 0
                except _lib.MagmaHipError as err:
                    print(f"[bench rank {rank}] RCCL communicator inside the library: {err}", file=sys.stderr)
                    status = 2 if "did not complete" in str(err) else 1
            worst = torch.tensor([status], device="cuda")
            dist.all_reduce(worst, op=dist.ReduceOp.MAX)
            if int(worst.item()) == 2:
                print(f"[bench rank {rank}] a collective of the self-check never completed; giving up", file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)
            if int(worst.item()) != 0:
                if e is not None:
                    e.close()
                e = None
        if e is None:
            from magma_amd.native import device_tensor
            if backend == "nccl":
                comm_kind = ("callback -> torch.distributed all_reduce (RCCL)" +
                             ("" if os.environ.get("MAGMA_BENCH_COMM", "rccl") != "rccl" else
                              "; the library's own communicator failed its self-check"))
            else:
                comm_kind = f"callback -> torch.distributed all_reduce ({backend}, staged through the host: rehearsal)"

            def allreduce(ptr, count, stream=None):
                torch.cuda.synchronize()
                tens = device_tensor(ptr, count)
                if backend == "nccl":
                    dist.all_reduce(tens)
                else:
                    host = tens.cpu()
                    dist.all_reduce(host)
                    tens.copy_(host)
                torch.cuda.synchronize()
            e = NativeEngine(dataset, rank=rank, world=world, allreduce=allreduce, device=dev)
            e.selfcheck(30000)
        return e
    eng = make_engine(ds)
    common = cfg["common_hp"]
    ini0, ini_i = np.array([1.0, 1.0]), np.tile([1.0, 1.0, 0.2], (ds.M, 1))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(th0, thi):
        n0, ni, l_old, l_new = eng.em_iteration(th0, thi, 0.0, common)
        if np.isnan(l_old):
            raise RuntimeError("EM iteration produced NaN hyper-parameters")
        return n0, ni, l_old, l_new

    theta0, theta_i = ini0.copy(), ini_i.copy()
    for _ in range(args.warmup):
        theta0, theta_i, _, _ = step(theta0, theta_i)
    if args.trajectory == "restart":
        theta0, theta_i = ini0.copy(), ini_i.copy()
    eng.reset_stats()
    lib.magma_prof_reset()
    lib.magma_prof_enable(1)
    lls, epss, step_ms = [], [], []
    fit = None
    barrier()
    t0 = time.perf_counter()
    if args.fit:
        trace = []
        theta0, theta_i, cv = eng.training(0.0, ini0, ini_i[0], common, trace=trace)
        lls = [r["logLL"] for r in trace]
        epss = [r["eps"] for r in trace]
        n_steps = len(trace)
    else:
        for _ in range(args.steps):
            t1 = time.perf_counter()
            theta0, theta_i, l_old, l_new = step(theta0, theta_i)
            step_ms.append(1e3 * (time.perf_counter() - t1))
            lls.append(l_old)
            epss.append((l_new - l_old) / abs(l_new))
        n_steps = args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    lib.magma_prof_enable(0)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    stats = eng.stats
    n_evals = stats["indiv_evals"]
    prof = {}
    for pid, name in enumerate(PROF_NAMES):
        cnt, ms = C.c_int(0), C.c_double(0.0)
        lib.magma_prof_get(pid, C.byref(cnt), C.byref(ms))
        if cnt.value:
            prof[name] = {"launches": cnt.value, "total_ms": ms.value, "avg_ms": ms.value / cnt.value}

    # the three rates behind an "EM iteration" (whose evaluation counts are data
    # dependent), each timed alone after the timed region on the final hyper-parameters
    def rate(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return reps / (time.perf_counter() - t1)
    pen_e = float(np.mean(theta_i[:, 2]))
    rates = {"e_step_per_s": rate(lambda: eng.e_step(theta0, theta_i, pen_e, 0.0))}
    # M-step alone (on the hyper-posterior the E-step above left resident), and its rate in sweeps:
    # one sweep = one f + grad f evaluation of every individual (theta_0 evaluations run next to them)
    eng.reset_stats()
    rates["m_step_per_s"] = rate(lambda: eng.m_step(theta0, theta_i, common))
    st_m = eng.stats
    sweeps = (st_m["common_evals"] if common else st_m["indiv_evals"] / max(1, eng.hi - eng.lo)) / 4.0
    rates["m_step_sweeps_per_s"] = rates["m_step_per_s"] * sweeps
    rates["m_step_counts"] = {"sweeps_per_m_step": sweeps, "theta0_evals_per_m_step": st_m["theta0_evals"] / 4.0}
    rates["logL_monitoring_per_s"] = rate(lambda: eng.logL_monitoring(theta0, theta_i))

    # BASELINE config 5 (M = 100,000 x 40 points, N = 2,000) sharded over the ranks of this job: the
    # scale-out workload, as an extra key next to the headline (which stays the weak-scaling c3 line
    # so that the per-N values the driver compares are one workload). A few EM iterations from the
    # initial hyper-parameters, timed like the headline.
    scale_out = None
    if world > 1 and args.config == "c3" and os.environ.get("MAGMA_BENCH_SCALE_OUT", "1") == "1":
        c5 = dict(CONFIGS["c5"])
        a5 = simu_arrays(int(os.environ.get("MAGMA_BENCH_SCALE_OUT_M", "100000")), c5["n"], c5["N"], False, False,
                         seed=c5["seed"])
        ds5 = Dataset.from_arrays(a5["off"], a5["t"], a5["y"], grid=a5["grid"])
        eng5 = make_engine(ds5)
        t5, ti5 = ini0.copy(), np.tile([1.0, 1.0, 0.2], (ds5.M, 1))
        t5, ti5, _, _ = eng5.em_iteration(t5, ti5, 0.0, False)
        t5, ti5 = ini0.copy(), np.tile([1.0, 1.0, 0.2], (ds5.M, 1))
        k5 = 3
        barrier()
        t1 = time.perf_counter()
        for _ in range(k5):
            t5, ti5, _, _ = eng5.em_iteration(t5, ti5, 0.0, False)
        barrier()
        el5 = time.perf_counter() - t1
        tt = torch.tensor([el5], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        scale_out = {"workload": f"c5: M={ds5.M} individuals x {c5['n']} points in total, sharded {world} ways "
                                 f"({eng5.hi - eng5.lo} per GPU), N={c5['N']}, individual hyper-parameters",
                     "scaling": "strong", "steps": k5, "ms_per_step": 1e3 * float(tt.item()) / k5,
                     "value": k5 * (ds5.M / 10000.0) / float(tt.item()), "unit": "EM it/s (10k-task units)"}
        eng5.close()

    if rank == 0:
        N, n, Mloc = cfg["N"], cfg["n"], eng.hi - eng.lo
        covfac = world == 1 and "cov_factor=0" not in os.environ.get("MAGMA_OPTIONS", "").replace(" ", "")
        units = M_total / 10000.0 if args.config in ("c3", "c5") else 1.0
        value = n_steps * units / elapsed
        S2 = Mloc * n ** 2
        nres = max(1, prof.get("indiv_mstep(resident)", {}).get("launches", 1))
        # algorithmic work per launch; FP64 FMA peak = FP64 MFMA peak on
        # MI355X, so the per-individual FP64 kernels are priced against the same 78.6 TFLOP/s
        alg = {
            # (with several ranks each forms its share of the block rows: priced on what runs)
            # one rank: Y = K^-1 V' with the triangular factor of cov (N^3 executed, the default);
            # several ranks or cov_factor=0: the full product T = K^-1 C (2 N^3, split over ranks)
            "gemm_nt(T=Kinv*C)": ("mfma", (1.0 if covfac else 2.0) * N ** 3 / world, True),
            # Z = T K^-1 is symmetric: only its lower tiles are formed (N^3 executed of the 2 N^3
            # a full product would take); priced on what runs
            "theta0_contract": ("mfma", 1.0 * N ** 3 / world, False),
            # one launch (the tile graph) up to 64 block rows; above that ~75 launches per call
            "spd_inverse": ("mfma", 1.0 * N ** 3, (N + 63) // 64 <= 64),
            # device-resident M-step: 5 n^3 per evaluation, all evaluations of a launch
            "indiv_mstep(resident)": ("mfma", 5.0 * n ** 3 * n_evals / nres, True),
            # E-step sums (three launches with the default deterministic route)
            "estep_scatter": ("hbm", 8.0 * S2 + 20.0 * Mloc * n, False),
        }
        for name, (bound, work, single) in alg.items():
            if name in prof:
                t_s = prof[name]["avg_ms"] * 1e-3
                ach = work / t_s / (1e12 if bound == "mfma" else 1e9)
                peak = PEAK_FP64_MFMA_TFLOPS if bound == "mfma" else PEAK_HBM_GBS
                prof[name].update({"bound": bound, "achieved": ach, "peak": peak,
                                   "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
                                   "frac": ach / peak, "single_kernel": single,
                                   "share_of_step_time": prof[name]["total_ms"] / (1e3 * elapsed)})
        # `roofline` = the single kernel with the largest HIP-event time in the timed region
        singles = [k for k in prof if alg.get(k, (0, 0, False))[2]]
        dom = max(singles, key=lambda k: prof[k]["total_ms"]) if singles else None
        roof = None
        if dom is not None:
            traffic, src = pmc_traffic(dom, args)
            roof = {"kernel": dom, "bound": prof[dom]["bound"], "achieved": prof[dom]["achieved"],
                    "peak": prof[dom]["peak"], "unit": prof[dom]["unit"], "frac": prof[dom]["frac"],
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": src,
                    "avg_launch_ms": prof[dom]["avg_ms"], "launches": prof[dom]["launches"],
                    "share_of_step_time": prof[dom]["share_of_step_time"]}
            if dom == "indiv_mstep(resident)":
                roof["n_evals_per_launch"] = n_evals / nres
                # the launch runs next to the theta_0 chain on its share of the chip (one workgroup
                # per compute unit, MAGMA_RESERVE_CUS left free): `frac` prices it against the
                # whole chip all the same
                if not common:
                    ncu = torch.cuda.get_device_properties(local).multi_processor_count
                    res = int(os.environ.get("MAGMA_RESERVE_CUS", 96))
                    if 0 < res < ncu:
                        roof["compute_units_used"] = ncu - res
                        roof["compute_units"] = ncu
                        roof["frac_of_peak_of_units_used"] = prof[dom]["frac"] * ncu / (ncu - res)
        e0 = stats["theta0_evals"] / n_steps
        ei = (stats["common_evals"] if common else stats["indiv_rounds"]) / n_steps
        # the fit inside the timed trajectory: iterations up to the reference's stopping rule
        # (eps < 1e-3, MAGMA.R:64-71), when the trajectory starts at the initial hyper-parameters
        if args.fit:
            fit = {"iterations": n_steps, "wall_ms": 1e3 * elapsed, "it_per_s": n_steps * units / elapsed,
                   "converged": bool(cv)}
        elif args.trajectory == "restart":
            stop = next((k for k, e in enumerate(epss) if e < 1e-3), None)
            if stop is not None:
                w = float(np.sum(step_ms[:stop + 1]))
                fit = {"iterations": stop + 1, "wall_ms": w, "it_per_s": (stop + 1) * units / (w * 1e-3),
                       "note": "EM iterations 1..k of the timed trajectory, k = first with eps < 1e-3"}
        steady = None
        if len(step_ms) >= 8:
            tail = step_ms[-5:]
            steady = {"ms_per_step": float(np.mean(tail)), "it_per_s": units / (np.mean(tail) * 1e-3),
                      "note": "last 5 timed iterations (past the stopping rule: cheaper M-steps)"}
        out = {
            "metric": "EM iterations/sec (E+M) at M=10k tasks", "value": value,
            "unit": "EM it/s (10k-task units)", "n_gpus": world, "steps": n_steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / n_steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (simu_scheme recipe, NumPy default_rng)",
            "config": {"workload": f"{args.config}: M={M_total} individuals x {n} points, union grid "
                                   f"N={N}, {'common' if common else 'individual'} hyper-parameters; "
                                   + ("one training() call to the stopping rule" if args.fit else
                                      f"{n_steps} EM iterations "
                                      + ("from the initial hyper-parameters" if args.trajectory == "restart"
                                         else "continuing after the warmup iterations")),
                       "M_per_gpu": Mloc, "theta0_evals_per_step": e0,
                       "indiv_eval_rounds_per_step": ei, "indiv_evals_per_step": n_evals / n_steps,
                       "parallelism": f"dp{world}", "communicator": comm_kind, "driver": "native (magma_em_iteration / magma_training)",
                       "logLL_trace": lls, "eps_trace": epss, "step_ms": step_ms},
            "roofline": roof, "fit": fit, "steady_state": steady, "rates": rates,
            "kernel_classes": prof,
        }
        if scale_out is not None:
            out["scale_out_c5"] = scale_out
        if not args.no_cpu_baseline and world == 1:
            t0c = time.time()
            cb = cpu_baseline(cfg)
            # one EM iteration of the reference at the evaluation counts measured above:
            # E-step + e0 theta_0 evaluations + ei sweeps over M individuals + 2 monitorings
            # (each ~ one theta_0 f-evaluation + one f-sweep; priced at half an f+g evaluation)
            t_iter = (cb["t_estep_dense"] + cb["t_estep_per_indiv"] * Mloc
                      + (e0 + 1.0) * cb["t_theta0_eval"]
                      + cb["t_indiv_eval"] * (n_evals / n_steps + Mloc))
            try:
                import threadpoolctl
                cores = max(i["num_threads"] for i in threadpoolctl.threadpool_info()) or 1
            except Exception:
                cores = os.cpu_count()
            out["cpu_baseline"] = {
                "value": units / t_iter,
                "unit": "EM it/s (10k-task units)", "cores": cores, "kind": "port",
                "sample": f"CPU restatement of the reference (R not present): dense parts at "
                          f"full N={N}; per-individual parts timed on "
                          f"{cb['Ms']} individuals and scaled to M={Mloc}; combined at the "
                          f"evaluation counts of the GPU run; {time.time() - t0c:.1f}s of CPU work",
                "pieces_s": {k: v for k, v in cb.items() if k != "Ms"}}
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
