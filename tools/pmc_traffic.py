#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected SEPARATELY -- they do not fit one TCC pass) into the
per-kernel HBM-traffic summary bench.py quotes as `roofline.traffic`.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-log-n 0
  python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_hbm_traffic_pmc.csv profiles/traffic_current.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): rocprofv3 reports both counters in KB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads as 64 bytes, so it is DOUBLED; WRITE_SIZE is taken as reported.
The JSON records the sha256 of the kernel sources the profile was taken on; bench.py ignores it when the sources have changed."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect(d, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ripp::", "")
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt


def main():
    fetch_dir, write_dir, out_csv, out_json = sys.argv[1:5]
    f, fc = collect(fetch_dir, "FETCH_SIZE"); w, wc = collect(write_dir, "WRITE_SIZE")
    from bench import csrc_sha256
    kernels = {}
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        per_launch = (2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024 / n
        kernels[k] = {"launches": n, "bytes_per_launch": per_launch, "fetch_KB_raw": f.get(k, 0.0), "write_KB": w.get(k, 0.0)}
        rows.append((k, n, f.get(k, 0.0), 2 * f.get(k, 0.0), w.get(k, 0.0), per_launch / 1e6))
    with open(out_csv, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, SEPARATE passes over the same command; KB as reported by rocprofv3;\n"
                "# fetch_KB_x2 applies the gfx950 correction of MI355X_MICROARCH.md (128-B requests of wide coalesced reads are tallied as 64 B).\n"
                "kernel,launches,fetch_KB_raw,fetch_KB_x2,write_KB,per_launch_MB_(x2_fetch+write)\n")
        for r in rows:
            o.write("%s,%d,%.0f,%.0f,%.0f,%.1f\n" % r)
    json.dump({"source": os.path.basename(out_csv) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)", "csrc_sha256": csrc_sha256(), "kernels": kernels},
              open(out_json, "w"), indent=1)
    for r in rows[:12]:
        print("%-40s launches %4d  per launch %10.1f MB" % (r[0][:40], r[1], r[5]))


if __name__ == "__main__":
    main()
