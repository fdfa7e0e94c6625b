"""Idle gaps between consecutive kernels of a single-stream rocprofv3 trace (rocpd .db): how much of the wall time is launch gap.
Usage: python tools/gap_summary.py trace.db [min_gap_us_to_list]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = sorted(db.execute(f"select start, end, {name_col} from kernels").fetchall())
busy = sum(e - s for s, e, _ in rows)
gaps = [(rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
small = [g for g in gaps if 0 <= g < 50_000]          # below 50 us: launch gaps (larger ones: host-side pauses between runs)
print(f"{len(rows)} kernels, busy {busy/1e6:.2f} ms, {len(small)} gaps < 50 us: total {sum(small)/1e6:.2f} ms, median {sorted(small)[len(small)//2]/1e3:.2f} us, "
      f"mean {sum(small)/len(small)/1e3:.2f} us; overlaps (negative gaps): {sum(1 for g in gaps if g < 0)}")
import collections
h = collections.Counter(min(int(g / 1000), 20) for g in small)
print("gap histogram (us: count):", dict(sorted(h.items())))
