#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/rpass.py textreact_amd/csrc/nn_ops.hip [name filter]
A kernel with ScratchSize > 0 spills; in the kernels that stage through LDS-DMA a spill reload waits behind the DMA queue."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
run = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra, capture_output=True, text=True)
out = run.stderr
if run.returncode != 0:
    sys.exit("hipcc failed:\n" + "\n".join(l for l in out.splitlines() if "error" in l)[:4000])
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[bytes/block\]| \[waves/SIMD\])?: (\d+)", l)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
bad = 0
for (k, v), name in zip(rows.items(), names):
    name = re.sub(r"^void \(anonymous namespace\)::|^void trx::", "", name)
    if flt in name:
        print("%-96s VGPR %3d AGPR %3d scratch %4d occ %d LDS %6d" % (name[:96], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1),
                                                                    v.get("Occupancy", -1), v.get("LDS Size", -1)))
        bad += v.get("ScratchSize", 0) > 0
print("%d kernels with scratch" % bad)
