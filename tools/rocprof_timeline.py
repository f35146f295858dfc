#!/usr/bin/env python3
"""The last N kernel dispatches of a rocprofv3 (rocpd sqlite) result, in launch order, with durations: which launch of a
kernel that runs several times per frame (the bounce levels) costs what.  usage: tools/rocprof_timeline.py results.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(db.execute("select name, start, duration from kernels order by start"))[-n:]
t0 = rows[0][1] if rows else 0
for name, start, dur in rows:
    short = name.replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{(start - t0) / 1e6:10.4f} ms  {dur / 1e6:9.4f} ms  {short}")
