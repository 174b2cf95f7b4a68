#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) as the `--stats` kernel table:
   name, calls, total/avg/min/max duration (ns), share.  Usage: rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                   f"max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by {name_col} "
                   f"order by 3 desc").fetchall() if {"vgpr_count", "lds_size"} <= set(cols) else \
    [r + (None, None, None, None) for r in cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
        f"group by {name_col} order by 3 desc").fetchall()]
tot = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS,Scratch"]
for r in rows:
    lines.append(f"\"{r[0][:90]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]}")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
