"""Summarise rocprofv3 --pmc passes (one counter per run, csv output) into the JSON bench.py reads.

usage: python profiles/pmc_summary.py out.json COUNTER=dir [COUNTER=dir ...]
Each dir holds the *_counter_collection.csv of one `rocprofv3 --pmc COUNTER` run. Values are
kept in the counter's own unit (FETCH_SIZE / WRITE_SIZE: KB); the gfx950 correction of the
guide's HBM section is applied by the reader (bench.py: pmc_traffic), not here.
"""
import csv
import glob
import json
import os
import re
import sys


def kernel_key(name):
    name = re.sub(r"^void\s+", "", name.strip())
    depth, out = 0, []
    for ch in name:  # cut the argument list, keep template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out)


def summarise(d, counter):
    per = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                key = kernel_key(row["Kernel_Name"])
                wg = max(int(row["Workgroup_Size"]), 1)
                per.setdefault(key, []).append((float(row["Counter_Value"]),
                                                int(row["Grid_Size"]) // wg))
    res = {}
    for key, vals in sorted(per.items()):
        mx, wgs = max(vals)
        res[key] = {"avg_kb": sum(v for v, _ in vals) / len(vals), "dispatches": len(vals),
                    "kb_per_workgroup_at_max": mx / max(wgs, 1), "max_kb": mx}
    return res


if __name__ == "__main__":
    out = {}
    for arg in sys.argv[2:]:
        counter, d = arg.split("=", 1)
        out[counter] = summarise(d, counter)
    with open(sys.argv[1], "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
