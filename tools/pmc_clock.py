#!/usr/bin/env python3
"""Achieved shader clock per kernel from a rocprofv3 pass with --pmc GRBM_GUI_ACTIVE --kernel-trace: busy cycles of the graphics engine during a
dispatch / the dispatch's duration.  Used for the one question DESIGN.md section 3 leaves open about the 104x line-buffer traffic of the pairing
kernels: does it cost CLOCK (power / DVFS headroom)?  Compare the pairing kernels' MHz with an ALU-only kernel of the same instruction mix
(tools/ubench/fqbench: carry-free Montgomery products in registers, no memory traffic) profiled the same way on the same box.

  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d D -o c --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0
  python3 tools/pmc_clock.py D [D2 ...]      # one table per directory
(The counter is summed over the XCDs by rocprofv3; the table divides by the number of XCD instances it finds, 8 on MI355X, when the raw figure
exceeds any plausible clock.)"""
import collections
import csv
import glob
import os
import sys


def table(d):
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    cyc = collections.defaultdict(float); ns = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ripp::", "")
            did = r["Dispatch_Id"]
            t = None
            if did in dur:
                t = dur[did][0]
            elif "Start_Timestamp" in r and r.get("End_Timestamp"):
                t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if t is None or t <= 0:
                continue
            cyc[k] += float(r["Counter_Value"]); ns[k] += t; cnt[k] += 1
    rows = []
    for k in sorted(cyc, key=lambda k: -ns[k])[:14]:
        mhz = cyc[k] / ns[k] * 1e3
        rows.append((k, cnt[k], ns[k] / 1e6, mhz / 8 if mhz > 4000 else mhz))
    return rows


if __name__ == "__main__":
    for d in sys.argv[1:]:
        print(f"# {d}: kernel, dispatches, total ms (under the profiler), achieved MHz = GRBM_GUI_ACTIVE / duration")
        for k, n, ms, mhz in table(d):
            print(f"{k[:48]:48s} {n:6d} {ms:10.2f} {mhz:8.0f}")
