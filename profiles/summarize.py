#!/usr/bin/env python3
"""Turn the rocprofv3 output of profiles/run_profile.sh <tag> (gpurun_out/prof_<tag>) into the committed summaries:
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json and profiles/traffic.json (what bench.py reports as
roofline.traffic).  FETCH_SIZE is doubled (gfx950 reads half of a wide coalesced stream, MI355X_MICROARCH.md, HBM);
Infinity-Cache hits are counted, so the figure is L2-miss traffic, an upper bound of the HBM bytes."""
import csv, glob, json, os, shutil, sys, collections

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)      # gpurun MERGES into gpurun_out/: take the newest run's file
assert stats, "no kernel_stats.csv under " + src


def head_is_what_was_profiled():
    """The summaries name a commit.  Refuse unless (1) the tracked sources are clean against HEAD and (2) every file the GPU
    box hashed when it took the profile (source.sha256, written by run_profile.sh) has the same hash in HEAD -- so
    traffic.json / <tag>_kernel_stats.csv can never describe another binary than the commit they cite."""
    import hashlib, subprocess
    dirty = subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "textreact_amd", "bench.py"]).decode().strip()
    if dirty:
        sys.exit("summarize.py: uncommitted changes under textreact_amd/ or bench.py:\n%s\ncommit first, then profile, then summarize" % dirty)
    f = os.path.join(src, "source.sha256")
    if not os.path.exists(f):
        sys.exit("summarize.py: %s missing (taken with an old run_profile.sh?)" % f)
    for line in open(f):
        want, rel = line.split()
        blob = subprocess.check_output(["git", "-C", root, "show", "HEAD:" + rel])
        if hashlib.sha256(blob).hexdigest() != want:
            sys.exit("summarize.py: %s changed between the profile run and HEAD; re-take the profile" % rel)
    return subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"]).decode().strip()


sha = head_is_what_was_profiled()
shutil.copy(stats[0], os.path.join(root, "profiles", tag + "_kernel_stats.csv"))


def is_boot(name):
    """knn_scan_kernel<L2, J, BOOT, NKS, RESCAN>: the third template argument marks the bootstrap launch, the fifth the
    re-scan of a batch's uncertified queries (usually none: a launch of a few microseconds) -- neither is the main scan"""
    args = [a.strip() for a in name.split("<", 1)[1].split(">", 1)[0].split(",")]
    return (len(args) > 2 and args[2] == "true") or (len(args) > 4 and args[4] == "true")


def counters(sub):
    agg = collections.defaultdict(list)
    name = None
    for f in sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)[:1]:   # newest run only
        for r in csv.DictReader(open(f)):
            if "knn_scan_kernel" in r["Kernel_Name"] and not is_boot(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                name = r["Kernel_Name"]
    return {k: sum(v) / len(v) for k, v in agg.items()}, name


avg_ms = None
for r in csv.DictReader(open(stats[0])):
    if "knn_scan_kernel" in r["Name"] and not is_boot(r["Name"]):
        avg_ms = float(r["AverageNs"]) / 1e6
        kname = r["Name"]
fetch, _ = counters("pmc_fetch")
write, _ = counters("pmc_write")
sq, _ = counters("pmc_sq")
fetch_kb, write_kb = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
sys.path.insert(0, root)
from bench import scan_source_sha256
out = {
    "git_sha": sha,
    "scan_source_sha256": scan_source_sha256(root),      # bench.py reports the traffic only for this kernel text
    "command": "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (profiles/run_profile.sh %s: one rocprofv3 pass for "
               "--kernel-trace --stats, one --pmc pass per counter group)" % tag,
    "kernel": kname, "avg_launch_ms_kernel_trace": avg_ms, "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE reads half of a wide (16 B/lane) coalesced stream -> doubled (MI355X_MICROARCH.md, HBM); "
                  "WRITE_SIZE exact; Infinity-Cache hits are counted, so this is L2-miss traffic = an upper bound of HBM bytes",
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 if fetch_kb is not None and write_kb is not None else None,
    "algorithmic_bytes_per_launch": 2 * (1000000 * 768 + 65536 * 768) + 65536 * 10 * 8,
    "sq": sq,
}
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
