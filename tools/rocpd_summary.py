"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel table (name, calls, total/avg/min/max us, %).
Usage: python tools/rocpd_summary.py gpurun_out/prof_xx/bench_results.db > profiles/rNN_kernel_stats.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"void |mvlpt::", "", short)
    a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f"# rocprofv3 --kernel-trace --stats summary ({len(rows)} dispatches, {tot/1e3:.2f} ms of kernel time)\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k[:110]}` | {a[0]} | {a[1]/1e3:.3f} | {a[1]/a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100*a[1]/tot:.1f} |")
