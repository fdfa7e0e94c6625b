"""MFMA-pipe utilisation and wave-state split per kernel from ONE rocprofv3 pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE \\
              SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -- python bench.py ...
Utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x clock x 1024 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are
quad-cycles (MI355X_MICROARCH.md).  The clock is taken as 2.0 GHz unless given (profiled passes run 1.9-2.0 GHz).
Usage: python tools/mfma_util.py <pmc.db> [clock_ghz] > profiles/rNN_gemm_pmc.md"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ghz = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cnt = {}
for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
    k = re.sub(r"\(.*", "", k)
    a = cnt.setdefault(k, {}).setdefault(c, [0, 0.0]); a[0] += 1; a[1] += float(v)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
dur = {}
for n, s, e in db.execute(f"select {name_col}, start, end from kernels"):
    a = dur.setdefault(re.sub(r"\(.*", "", n), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
print(f"# MFMA utilisation per kernel (rocprofv3 --pmc SQ_* pass, clock assumed {ghz} GHz, 1024 SIMDs)\n")
print("| kernel | launches | avg us | MFMA busy % of SIMD-cycles | waves parked (WAIT_ANY) % | issue-stalled (WAIT_INST_ANY) % | issuing % | LDS active % of CU-cycles | LDS bank-conflict cycles |")
print("|---|---|---|---|---|---|---|---|---|")
rows = []
for k, c in cnt.items():
    if k not in dur or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    n, us = dur[k][0], dur[k][1] / dur[k][0]
    avg = lambda x: c[x][1] / c[x][0] if x in c else float("nan")
    cyc = us * 1e-6 * ghz * 1e9
    wc = avg("SQ_WAVE_CYCLES")
    rows.append((us * n, k, n, us, 100 * avg("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024), 100 * avg("SQ_WAIT_ANY") / wc, 100 * avg("SQ_WAIT_INST_ANY") / wc,
                 100 * avg("SQ_ACTIVE_INST_ANY") / wc, 100 * avg("SQ_LDS_IDX_ACTIVE") / (cyc * 256), avg("SQ_LDS_BANK_CONFLICT")))
for _, k, n, us, mf, wa, wi, ai, lds, bc in sorted(rows, reverse=True)[:14]:
    print(f"| `{k[:90]}` | {n} | {us:.1f} | {mf:.1f} | {wa:.1f} | {wi:.1f} | {ai:.1f} | {lds:.1f} | {bc:.0f} |")
