#!/usr/bin/env python3
"""gpurun_out/clock_<tag> (profiles/run_clock.sh) -> profiles/<tag>_clock.json: the effective clock of the production scan
kernel (GRBM_GUI_ACTIVE / 8 / dispatch duration) and the in-kernel clocks of the contraction lab with its parts switched on
one by one.  The numbers DESIGN.md 3.1 rests its "power-limited, not issue-limited" reading on."""
import csv, glob, json, os, re, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "clock_" + tag)
disp = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_grbm", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime, reverse=True)[:1]:   # newest run only
    for r in csv.DictReader(open(f)):
        disp[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
clocks = []
for f in sorted(glob.glob(os.path.join(src, "pmc_grbm", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)[:1]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "knn_scan_kernel" in name and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            args = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
            if len(args) > 2 and args[2] == "true":
                continue                    # the bootstrap launch
            ns = disp.get(r["Dispatch_Id"], (None, None))[1]
            if ns:
                clocks.append({"ghz": float(r["Counter_Value"]) / 8.0 / ns, "ms": ns / 1e6})
clocks.sort(key=lambda c: c["ghz"])
lab = {}
names = {"lab_v4_f9": "lab variant 4, MFMAs on register operands (no LDS reads, no DMA)", "lab_v4_f1": "lab variant 4, + LDS fragment reads",
         "lab_v4_f8": "lab variant 4, + LDS-DMA fill, no fragment reads", "lab_v4_f0": "lab variant 4, everything",
         "lab_v2_f1": "lab variant 2 (the production loop), MFMAs + LDS fragment reads, no DMA", "lab_v2_f0": "lab variant 2 (the production loop), everything"}
for key, what in names.items():
    p = os.path.join(src, key + ".log")
    if not os.path.exists(p):
        continue
    t = open(p).read()
    runs = re.findall(r": ([0-9.]+) ms\s+([0-9.]+) TFLOP/s", t)
    m1 = re.match(r"(.*) (.*)", "%s %s" % min(runs, key=lambda r: float(r[0]))) if runs else None      # the first launch set is cold
    m2 = re.search(r"in-kernel clock: median ([0-9.]+) GHz \(min ([0-9.]+), max ([0-9.]+)\); cycles per K-step ([0-9.]+)", t)
    if m1 and m2:
        lab[key] = {"what": what, "ms": float(m1.group(1)), "tflops": float(m1.group(2)), "clock_ghz_median": float(m2.group(1)),
                    "clock_ghz_min": float(m2.group(2)), "clock_ghz_max": float(m2.group(3)), "cycles_per_kstep": float(m2.group(4))}
try:       # the commit the GPU run was taken at: second argument, else HEAD
    sha = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    sha = "unknown"
out = {"git_sha": sha,
       "how": "profiles/run_clock.sh %s: (1) rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace over bench.py --steps 10: effective clock = "
              "GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration; (2) tools/scan_lab, 30 back-to-back launches of the C1 contraction on random "
              "data: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz, median over workgroups" % tag,
       "production_scan_kernel": {"launches": len(clocks),
                                  "effective_clock_ghz_median": clocks[len(clocks) // 2]["ghz"] if clocks else None,
                                  "effective_clock_ghz_min": clocks[0]["ghz"] if clocks else None,
                                  "effective_clock_ghz_max": clocks[-1]["ghz"] if clocks else None,
                                  "launch_ms_median": sorted(c["ms"] for c in clocks)[len(clocks) // 2] if clocks else None,
                                  "max_clock_ghz": 2.4},
       "lab": lab}
json.dump(out, open(os.path.join(root, "profiles", tag + "_clock.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
