"""Per-queue view of a multi-stream rocprofv3 kernel trace (rocpd .db): for every HW queue the busy time, the idle gaps between
its consecutive kernels and the largest kernel classes, inside the steady-state window [t0 + skip, t1 - tail] — who waits for whom in
the overlapped step.  Usage: python tools/stream_timeline.py trace.db [skip_fraction]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select start, end, {name_col}, {qcol} from kernels order by start").fetchall()
# steady state: between the 30th and 70th percentile (by start time) of the busiest queue's kernels
busiest = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
st = sorted(r[0] for r in rows if r[3] == busiest)
lo, hi = st[int(0.3 * len(st))], st[int(0.7 * len(st))]
rows = [r for r in rows if lo <= r[0] < hi]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
print(f"window {span:.2f} ms, {len(rows)} kernels, queues: {sorted(set(r[3] for r in rows))}")
byq = collections.defaultdict(list)
for s, e, n, q in rows:
    byq[q].append((s, e, re.sub(r"\(.*", "", n)))
for q, ks in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in ks) / 1e6
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    small = sum(g for g in pos if g < 20_000) / 1e6
    mid = sum(g for g in pos if 20_000 <= g < 500_000) / 1e6
    big = sum(g for g in pos if g >= 500_000) / 1e6
    top = collections.Counter()
    for s, e, n in ks:
        top[n[-60:]] += (e - s) / 1e6
    print(f"queue {q}: {len(ks)} kernels, busy {busy:.2f} ms ({100 * busy / span:.0f} % of the window); gaps < 20 us: {small:.2f} ms, 20-500 us: {mid:.2f} ms, >= 500 us: {big:.2f} ms")
    for n, v in top.most_common(4):
        print(f"      {v:8.2f} ms  {n}")
