#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / average / min / max duration,
and, when the run collected counters (--pmc), the per-kernel sum of every counter.
usage: tools/rocprof_summary.py results.db [> profiles/NAME.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel summary of {path}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{name[:60]:60s} {n:6d} {tot / 1e6:10.3f} {avg / 1e6:10.4f} {mn / 1e6:10.4f} {mx / 1e6:10.4f} {100 * tot / total:6.2f}")
    try:
        # one row per (dispatch, counter, dimension instance): sum the instances, then average over dispatches
        pm = list(c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                            "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"))
    except sqlite3.Error as e:
        pm = []
        print(f"# (no counters: {e})")
    if pm:
        print("\n# counters: kernel, counter, dispatches, sum over dispatches, per dispatch")
        for name, cn, n, v in pm:
            print(f"{name[:60]:60s} {cn:28s} {n:6d} {v:20.0f} {v / n:20.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
