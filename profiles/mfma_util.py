"""FP64 MFMA utilisation per kernel from one profile_round.sh output directory.

usage: python profiles/mfma_util.py gpurun_out/prof_<tag>_<cfg> out.json [cfg steps warmup]
Joins the PMC pass (SQ_INSTS_VALU_MFMA_MOPS_F64: 1 MOP = 512 flops; SQ_VALU_MFMA_BUSY_CYCLES: a
v_mfma_f64_16x16x4 = 4 MOPs = 64 busy cycles) with the kernel trace's durations of the same
command line (kernel_stats.csv)."""
import csv
import json
import sys

from pmc_summary import kernel_key

d, out = sys.argv[1], sys.argv[2]
cfg, steps, warm = (sys.argv[3:6] + ["c3", "20", "5"][len(sys.argv[3:6]):])
pmc = json.load(open(f"{d}/pmc_{cfg}_{steps}_{warm}.json"))
stats = {kernel_key(r["Name"]): r for r in csv.DictReader(open(f"{d}/kernel_stats.csv"))}
res = {}
for k, mops in pmc["SQ_INSTS_VALU_MFMA_MOPS_F64"].items():
    if k not in stats or mops["avg_kb"] == 0:
        continue
    s = stats[k]
    avg_ms = float(s["AverageNs"]) * 1e-6
    flops = mops["avg_kb"] * 512.0
    busy = pmc["SQ_VALU_MFMA_BUSY_CYCLES"][k]["avg_kb"]
    res[k] = {"launches": int(s["Calls"]), "avg_duration_ms": avg_ms, "min_ms": float(s["MinNs"]) * 1e-6,
              "max_ms": float(s["MaxNs"]) * 1e-6, "mfma_flops_executed_per_launch": flops,
              "executed_tflops": flops / (avg_ms * 1e-3) / 1e12,
              "frac_of_fp64_peak_78.6": flops / (avg_ms * 1e-3) / 78.6e12,
              "mfma_busy_cycles_per_launch": busy,
              "mfma_busy_frac_of_1024_simds_at_2.4GHz": busy / (1024 * 2.4e9 * avg_ms * 1e-3)}
json.dump({"_note": __doc__.split("usage")[0].strip() + f" Command: bench.py --config {cfg} --steps {steps} --warmup {warm}.",
           "kernels": res}, open(out, "w"), indent=1, sort_keys=True)
