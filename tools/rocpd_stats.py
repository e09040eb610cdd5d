"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / avg / min / max (us)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name)
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"{'kernel':60s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:60]:60s} {a[0]:7d} {a[1]/1e3:12.1f} {a[1]/a[0]/1e3:10.2f} {a[2]/1e3:10.2f} {a[3]/1e3:10.2f} {100*a[1]/tot:6.2f}")
print(f"total kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches")
