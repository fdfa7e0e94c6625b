"""Per-kernel PMC counter averages from a rocprofv3 rocpd .db (counters_collection / pmc_events views)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
def cols(t): return [r[1] for r in db.execute(f"pragma table_info({t})")]
cc = cols("counters_collection")
print("# columns:", cc, file=sys.stderr)
rows = db.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cc)}
kname = [c for c in cc if "kernel" in c and "name" in c] or [c for c in cc if c == "name"]
agg = {}
for r in rows:
    k = re.sub(r"\(.*", "", str(r[ix[kname[0]]]))[:70]
    cname = r[ix["counter_name"]] if "counter_name" in ix else r[ix["name"]]
    val = r[ix["value"]] if "value" in ix else r[ix["counter_value"]]
    a = agg.setdefault((k, cname), [0, 0.0]); a[0] += 1; a[1] += float(val)
ks = sorted({k for k, _ in agg})
cs = sorted({c for _, c in agg})
for k in ks:
    print(k)
    for c in cs:
        if (k, c) in agg:
            n, v = agg[(k, c)]
            print(f"    {c:32s} avg/dispatch {v/n:16.1f}  (n={n})")
