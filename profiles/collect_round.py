"""Copy the summaries of profile_round.sh runs from gpurun_out/prof_<tag>_<cfg>/ into profiles/
(tracked): python profiles/collect_round.py r03 c3:20:5 c2:5:2 c4:5:2 c5:5:2"""
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
for spec in sys.argv[2:]:
    cfg, steps, warm = spec.split(":")
    src = os.path.join(root, "gpurun_out", f"prof_{tag}_{cfg}")
    dst = os.path.join(root, "profiles")
    stem = f"{tag}_bench_{cfg}_{steps}_{warm}"
    pairs = [("bench.json", f"{stem}.json"), ("bench_under_rocprof.json", f"{stem}_under_rocprof.json"),
             ("kernel_stats.csv", f"{stem}_kernel_stats.csv"),
             (f"pmc_{cfg}_{steps}_{warm}.json", f"{tag}_pmc_{cfg}_{steps}_{warm}.json"),
             ("mfma_util.json", f"{tag}_mfma_util_{cfg}.json"),
             ("timeline_compact.csv", f"{tag}_timeline_{cfg}_{steps}_{warm}.csv")]
    for a, b in pairs:
        p = os.path.join(src, a)
        if os.path.exists(p) and os.path.getsize(p) < (3 << 20):
            shutil.copyfile(p, os.path.join(dst, b))
            print("copied", b)
        else:
            print("skipped", a)
    if cfg == "c3":  # the name VERDICT r02 asked for
        shutil.copyfile(os.path.join(src, "mfma_util.json"), os.path.join(dst, f"{tag}_mfma_util.json"))
