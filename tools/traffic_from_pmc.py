"""Average HBM-side traffic per gemm_bt launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md §HBM prescribes: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads
(all reads of this kernel are 16 B/lane) -> doubled; WRITE_SIZE is taken as is (it matches the known output sizes of
the LayerNorm / residual-GEMM launches exactly).  Units of the raw counters: KB.
Usage: python tools/traffic_from_pmc.py <fetch.db> <write.db> [label] > profiles/gemm_hbm_traffic.json"""
import json, re, sqlite3, sys

def per_kernel(dbp, counter):
    db = sqlite3.connect(dbp)
    agg = {}
    for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if c != counter: continue
        a = agg.setdefault(re.sub(r"\(.*", "", k), [0, 0.0]); a[0] += 1; a[1] += float(v)
    return agg
rd, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
g = [k for k in rd if ("gemm_bt_kernel" in k or "gemm_bt_phased_kernel" in k or "gemm_pc_kernel" in k or "gemm_pcp_kernel" in k) and "sgemm" not in k]
n = sum(rd[k][0] for k in g); fr = sum(rd[k][1] for k in g)
nw = sum(wr[k][0] for k in g if k in wr); fw = sum(wr[k][1] for k in g if k in wr)
what = sys.argv[3] if len(sys.argv) > 3 else "headline bench step"
# identity of the binary the counters were taken on (mvlpt_version(): "... src:<hash> git:<commit>"); bench.py drops the figure when
# the library it loads is another one
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from mvlpt_amd import _lib
    version = _lib.lib.mvlpt_version().decode()
except Exception as e:      # noqa: BLE001
    version = f"unknown ({e})"
out = {"lib_version": version, "lib_src_hash": (re.search(r"src:(\w+)", version) or [None, None])[1],
       "kernel": f"gemm_bt_kernel + gemm_bt_phased_kernel + gemm_pc_kernel + gemm_pcp_kernel (all epilogues/geometries, {what})", "launches_sampled": n,
       "FETCH_SIZE_kb_avg_raw": round(fr / n, 1), "WRITE_SIZE_kb_avg_raw": round(fw / nw, 1),
       "bytes_per_launch": int((2.0 * fr / n + fw / nw) * 1024),
       "correction": "2 x FETCH_SIZE (gfx950 half-count of 16 B/lane reads) + WRITE_SIZE; counters in KB; includes Infinity-Cache hits",
       "source": "two separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace / rocprofv3 --pmc WRITE_SIZE --kernel-trace "
                 "-- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing"}
# per instantiation (same correction), largest first: where the bytes of the average launch come from
rows = []
for k in g:
    nk_, wk = rd[k][0], wr.get(k, [1, 0.0])
    rows.append({"kernel": re.sub(r"^_ZN5mvlpt\d+", "", k)[:64], "launches": nk_, "MB_per_launch": round((2.0 * rd[k][1] / nk_ + wk[1] / max(wk[0], 1)) / 1024, 1),
                 "fetch_MB_raw": round(rd[k][1] / nk_ / 1024, 1), "write_MB": round(wk[1] / max(wk[0], 1) / 1024, 1)})
out["per_kernel"] = sorted(rows, key=lambda r: -r["MB_per_launch"] * r["launches"])
print(json.dumps(out, indent=1))
