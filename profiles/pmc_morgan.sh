#!/bin/bash
# SQ counters of the scan kernel's fp4 and int8 forms on the morgan workload (400,000 x 1024 bit vectors searching themselves):
#   gpurun -- bash profiles/pmc_morgan.sh   -> one JSON line (profiles/r04_morgan_pmc.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_morgan
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for form in fp4 i8; do
  if [ $form = i8 ]; then export TRX_NO_FP4=1; fi
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/$form -- python3 $R/bench.py --workload morgan --n-corpus 400000 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$form.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
out={}
for form in ("fp4","i8"):
    f=glob.glob("$OUT/%s/*/*_counter_collection.csv"%form)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "knn_scan_kernel" in n and ", false, 0, false, %s>"%("2" if form=="fp4" else "1") in n:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    m["launches"]=len(next(iter(agg.values()))) if agg else 0
    if m.get("SQ_BUSY_CYCLES"):      # (same aggregation as profiles/r04_pmc_summary.json: 25.8 there for the bf16 headline scan at 0.58 of its peak)
        m["SQ_VALU_MFMA_BUSY_CYCLES_per_SQ_BUSY_CYCLE"]=m["SQ_VALU_MFMA_BUSY_CYCLES"]/m["SQ_BUSY_CYCLES"]
    out[form]=m
print(json.dumps(out))
PY
