#!/usr/bin/env python3
"""gpurun_out/forms_<tag> (profiles/pmc_forms.sh) -> profiles/traffic_forms.json: per workload of the reference (fingerprint = the
int8 form of the scan, morgan = its fp4 form) the main scan launches' average duration and L2-miss bytes per launch.  FETCH_SIZE
is doubled (gfx950 reads half of a wide coalesced stream: MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted, so the
figure is an upper bound of the HBM bytes.  Tied to the scan kernel's source text like profiles/traffic.json: bench.py reports
roofline.traffic for these workloads only when the hash matches what it compiled from."""
import csv, glob, hashlib, json, os, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "forms_" + tag)
h = hashlib.sha256()
for rel in ("textreact_amd/csrc/knn_scan.hip", "textreact_amd/csrc/knn_common.h"):
    blob = open(os.path.join(root, rel), "rb").read()
    h.update(blob)
    want = [l.split()[0] for l in open(os.path.join(src, "source.sha256")) if l.split()[1] == rel][0]
    if hashlib.sha256(blob).hexdigest() != want:
        sys.exit("summarize_forms.py: %s changed since the profile was taken; re-take it" % rel)


def is_main(name, fmt):      # knn_scan_kernel<L2, J, BOOT, NKS, RESCAN, FMT>: not the bootstrap, not the re-scan, the wanted form
    if "knn_scan_kernel" not in name:
        return False
    a = [x.strip() for x in name.split("<", 1)[1].split(">", 1)[0].split(",")]
    return a[2] == "false" and a[4] == "false" and a[5] == str(fmt)


out = {"scan_source_sha256": h.hexdigest(), "tag": tag,
       "correction": "FETCH_SIZE x 2 (gfx950 wide-stream correction), WRITE_SIZE exact, both in KiB; Infinity-Cache hits counted: an upper bound of the HBM bytes"}
for wl, fmt in (("fingerprint", 1), ("morgan", 2)):
    res = {"form": {1: "int8 (v_mfma_i32_16x16x64_i8)", 2: "fp4 (v_mfma_scale_f32_16x16x128_f8f6f4)"}[fmt]}
    f = sorted(glob.glob(os.path.join(src, wl + "_trace", "**", "*kernel_stats.csv"), recursive=True))
    for r in csv.DictReader(open(f[-1])):
        if is_main(r["Name"], fmt):
            res["kernel"] = r["Name"]; res["launches"] = int(r["Calls"]); res["avg_launch_ms_kernel_trace"] = float(r["AverageNs"]) / 1e6
    for sub, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        vals = []
        f = sorted(glob.glob(os.path.join(src, wl + "_" + sub, "**", "*counter_collection.csv"), recursive=True))
        for r in csv.DictReader(open(f[-1])):
            if r["Counter_Name"] == key and is_main(r["Kernel_Name"], fmt):
                vals.append(float(r["Counter_Value"]))
        # (the last launch of a self-search covers the remainder of the queries: average over the full 65,536-query launches)
        full = sorted(vals)[len(vals) // 4:] if len(vals) > 4 else vals
        res[key + "_KB"] = sum(full) / max(len(full), 1)
    res["hbm_bytes_per_launch"] = res["FETCH_SIZE_KB"] * 1024 * 2 + res["WRITE_SIZE_KB"] * 1024
    out[wl] = res
json.dump(out, open(os.path.join(root, "profiles", "traffic_forms.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
