#!/usr/bin/env python3
"""gpurun_out/prof_<tag>_predictor (profiles/run_profile_predictor.sh) -> profiles/<tag>_predictor_pmc.json: per kernel and
shape the median launch duration, achieved TFLOP/s or GB/s against the peak, matrix-core and vector busy fractions
(SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES-equivalent), SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES) and the HBM-side
traffic (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md HBM section) next to the algorithmic bytes."""
import collections, csv, glob, json, os, statistics, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_%s_predictor" % tag)
plan = json.loads(open(os.path.join(src, "plan.json")).read().split("PLAN ", 1)[1])
N = plan["N"]


def dispatches(sub):
    """per kernel-name substring: list of dispatches in launch order, each {counter: value, 'us': duration}"""
    f = sorted(glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)      # gpurun MERGES into gpurun_out/: take the newest run's file
    by = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        key = (r["Dispatch_Id"], r["Kernel_Name"])
        d = by.setdefault(key, {"name": r["Kernel_Name"], "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    return sorted(by.values(), key=lambda d: 0)  # csv is in dispatch order already


def blocks(ds, substr, nshape, per_call=1):
    sel = [d for d in ds if substr in d["name"]]
    n = N * per_call
    assert len(sel) >= nshape * n, (substr, len(sel), nshape * n)
    return [sel[i * n:(i + 1) * n] for i in range(nshape)]


def med(block, key, skip=2):
    vals = [d[key] for d in block[skip:] if key in d]
    return statistics.median(vals) if vals else None


sq, fe, wr = dispatches("pmc_sq"), dispatches("pmc_fetch"), dispatches("pmc_write")
out = {"command": "python3 tools/predictor_shapes.py %d under rocprofv3 (profiles/run_profile_predictor.sh %s)" % (N, tag), "kernels": []}
KERN = [("attention_fwd_mfma_kernel", "attention", "flops_fwd", 1.0), ("attention_bwd_dq_mfma_kernel", "attention", "flops_fwd", 1.5),
        ("attention_bwd_dkv_mfma_kernel", "attention", "flops_fwd", 2.0), ("add_ln_fwd_vec_kernel", "add_ln", "bytes_fwd", 1.0),
        ("add_ln_bwd_vec_kernel", "add_ln", "bytes_bwd", 1.0),
        ("gemm_tn_kernel<true>", "gemm_tn_grouped", "flops", 1.0), ("gemm_tn_kernel<false>", "gemm_tn_split", "flops", 1.0)]
for kname, fam, workkey, mult in KERN:
    shapes = plan[fam]
    try:
        bsq, bfe, bwr = blocks(sq, kname, len(shapes)), blocks(fe, kname, len(shapes)), blocks(wr, kname, len(shapes))
    except AssertionError as e:
        out["kernels"].append({"kernel": kname, "error": str(e)})
        continue
    for i, sh in enumerate(shapes):
        us = med(bsq[i], "us")
        rec = {"kernel": kname, "shape": sh["shape"], "median_us": us}
        work = sh[workkey] * mult
        if fam.startswith("gemm_tn"):       # the weight-gradient GEMM: MFMA-bound (2 M N K), its bytes beside it
            rec["achieved_TFLOPs"] = work / (us * 1e-6) / 1e12
            rec["frac_of_2500_TFLOPs"] = rec["achieved_TFLOPs"] / 2500.0
            rec["algorithmic_bytes"] = sh["bytes"]
        elif fam == "attention":       # algorithmic FLOPs: 4 B H Lq Lk 64 forward; dq pass 6/4, dk/dv pass 8/4 of it (GEMM units)
            rec["achieved_TFLOPs"] = work / (us * 1e-6) / 1e12
            rec["frac_of_2500_TFLOPs"] = rec["achieved_TFLOPs"] / 2500.0
            bkey = {"attention_fwd_mfma_kernel": "bytes_fwd", "attention_bwd_dq_mfma_kernel": "bytes_dq", "attention_bwd_dkv_mfma_kernel": "bytes_dkv"}[kname]
            if bkey in sh:       # the 7-position decoder shapes are byte-bound: K and V of 512 keys against 7 queries
                rec["algorithmic_bytes"] = sh[bkey]
                rec["frac_of_8000_GBs"] = sh[bkey] / (us * 1e-6) / 1e9 / 8000.0
        else:
            rec["algorithmic_bytes"] = work
            rec["achieved_GBs"] = work / (us * 1e-6) / 1e9
            rec["frac_of_8000_GBs"] = rec["achieved_GBs"] / 8000.0
        busy, wavec = med(bsq[i], "SQ_BUSY_CYCLES"), med(bsq[i], "SQ_WAVE_CYCLES")
        mfma, valu = med(bsq[i], "SQ_VALU_MFMA_BUSY_CYCLES"), med(bsq[i], "SQ_ACTIVE_INST_VALU")
        gui = med(bsq[i], "GRBM_GUI_ACTIVE")
        rec["SQ_WAVES"] = med(bsq[i], "SQ_WAVES")
        if gui and mfma is not None:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; the matrix pipes of the chip: 1024 SIMDs
            rec["mfma_busy_frac_of_simd_cycles"] = mfma / (gui / 8.0 * 1024.0)
        if wavec and valu is not None:
            rec["valu_active_frac_of_wave_cycles"] = valu / wavec      # both count quad-cycles
        f_kb, w_kb = med(bfe[i], "FETCH_SIZE"), med(bwr[i], "WRITE_SIZE")
        if f_kb is not None and w_kb is not None:
            rec["hbm_side_bytes"] = (2 * f_kb + w_kb) * 1024
        out["kernels"].append(rec)
json.dump(out, open(os.path.join(root, "profiles", tag + "_predictor_pmc.json"), "w"), indent=1)
for r in out["kernels"]:
    print(json.dumps(r))
