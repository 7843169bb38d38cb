#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the engine as compiled for gfx950 (hipcc -Rpass-analysis=kernel-resource-usage).
scratch = the private segment per lane: spilled registers (column `spills`, in dwords) PLUS the stack frames of the out-of-line complete-formula
fallbacks a kernel calls for exceptional lanes (cold: a kernel with spills = 0 touches no scratch on ordinary inputs).
  python tools/kernel_resources.py [-DRIPP_BLS12_377]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ripp_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-o", "/dev/null", "engine.hip"] + sys.argv[1:]
err = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"remark: (?:[^ ]*:\d+:\d+: )?(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
print("%-64s %5s %5s %8s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "spills", "occ", "LDS"))
for (k, v), name in zip(rows.items(), names):
    name = re.sub(r"\(.*", "", name).replace("ripp::", "").replace("void ", "")
    print("%-64s %5s %5s %8s %6s %4s %7s" % (name[:64], v.get("VGPRs"), v.get("AGPRs"), v.get("ScratchSize [bytes/lane]"), v.get("VGPRs Spill"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
