"""Per-kernel counter averages from a rocprofv3 rocpd database (--pmc run): python tools/pmc_db.py <results.db or dir> [kernel name filter]"""
import collections
import glob
import os
import sqlite3
import sys

path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)
for db in dbs:
    con = sqlite3.connect(db)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, c, v, d in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        if flt in k:
            agg[k][c].append(v)
            agg[k]["__duration_ns"].append(d)
    for k, cs in agg.items():
        print(k[:120])
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        for c in sorted(m):
            print(f"   {c:28s} {m[c]:16.1f}  ({len(cs[c])} samples)")
        if "SQ_INSTS_MFMA" in m and m["SQ_INSTS_MFMA"] > 0:
            print(f"   VALU per MFMA              {m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']:8.2f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
            print(f"   MFMA busy                  {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):8.3f}")
