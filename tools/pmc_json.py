#!/usr/bin/env python3
"""Merge the rocprofv3 passes of tools/profile_round.sh into the JSON bench.py reads (profiles/rNN_pmc.json): per
kernel and per launch, the average duration, SQ_INSTS_VALU, SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, the wave-cycle
counters, GRBM_GUI_ACTIVE (-> effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration), FETCH_SIZE and WRITE_SIZE in KiB
and, from the instruction-class passes, every SQ_INSTS_VALU_* class (FP32 / FP64 add, mul, fma, transcendental, INT32,
INT64, conversions) that bench.py weights with the measured issue costs.
usage: tools/pmc_json.py out.json workload_key "command" sq.db rd.db wr.db [class.db ...]"""
import json
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name


def read(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    dur = {short(n): (cnt, avg) for n, cnt, avg in c.execute("select name, count(*), avg(duration) from kernels group by name")}
    ctr = {}
    for name, cn, nd, v in c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                                     "from counters_collection group by kernel_name, counter_name"):
        ctr.setdefault(short(name), {})[cn] = v / nd
    return dur, ctr


def main(out, key, command, sq, rd, wr, *classes):
    kernels = {}
    dur, ctr = read(sq)
    for k, c in ctr.items():
        if not k.startswith("k_"):
            continue
        e = kernels.setdefault(k, {})
        e["launches_profiled"], e["duration_ms"] = dur[k][0], dur[k][1] / 1e6
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                     "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if name in c:
                e[name] = c[name]
        if "GRBM_GUI_ACTIVE" in c:
            e["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8 / dur[k][1]
    for path, name in ((rd, "FETCH_SIZE"), (wr, "WRITE_SIZE")):
        _, ctr = read(path)
        for k, c in ctr.items():
            if k in kernels and name in c:
                kernels[k][name + "_KiB"] = c[name]
    for path in classes:
        _, ctr = read(path)
        for k, c in ctr.items():
            if k in kernels:
                for name, v in c.items():
                    if name.startswith("SQ_INSTS_VALU_"):
                        kernels[k][name] = v
    json.dump({"workload_key": key, "command": command,
               "source": "rocprofv3 --kernel-trace --pmc, one run per counter group (tools/profile_round.sh); values are per launch, "
                         "averaged over the profiled launches; summaries of the same runs: profiles/rNN_pmc_{sq,rd,wr,fl,fl2}.txt",
               "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)
    print(open(out).read())


if __name__ == "__main__":
    main(*sys.argv[1:])
