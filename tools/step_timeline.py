"""One steady-state step of the overlapped headline loop from a rocprofv3 kernel trace: per queue, when it starts, ends and how busy it
is between two consecutive logits kernels.  Usage: python tools/step_timeline.py trace.db [which_step]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute(f"select start, end, {name_col}, queue_id from kernels order by start").fetchall()
marks = [r[0] for r in rows if "logits_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[k], marks[k + 1]
print(f"step {k}: {(b - a) / 1e6:.3f} ms between two logits kernels")
sel = [r for r in rows if a <= r[0] < b]
for q in sorted(set(r[3] for r in sel)):
    ks = [r for r in sel if r[3] == q]
    busy = sum(e - s for s, e, _, _ in ks) / 1e6
    print(f"queue {q}: {len(ks):4d} kernels, first start +{(ks[0][0] - a) / 1e6:6.3f} ms, last end +{(max(r[1] for r in ks) - a) / 1e6:6.3f} ms, busy {busy:6.3f} ms")
    # coarse phases: a new phase starts after an idle gap of more than 100 us
    ph_s, ph_e, n = ks[0][0], ks[0][1], 1
    for s, e, nm, _ in ks[1:]:
        if s - ph_e > 100_000:
            print(f"      +{(ph_s - a) / 1e6:6.3f} .. +{(ph_e - a) / 1e6:6.3f} ms ({n} kernels)")
            ph_s, n = s, 0
        ph_e = max(ph_e, e); n += 1
    print(f"      +{(ph_s - a) / 1e6:6.3f} .. +{(ph_e - a) / 1e6:6.3f} ms ({n} kernels)")
if len(sys.argv) > 3:      # list the first kernels of one queue in that step: python tools/step_timeline.py db step queue [count]
    q, cnt = int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 14
    t_from = float(sys.argv[5]) if len(sys.argv) > 5 else -1.0          # optional: only kernels starting after +t_from ms
    for s, e, nm, _ in [r for r in sel if r[3] == q and (r[0] - a) / 1e6 >= t_from][:cnt]:
        short = re.sub(r"\(.*", "", nm)[-70:]
        print(f"   +{(s - a) / 1e6:7.3f} ms  {(e - s) / 1e3:7.1f} us  {short}")
