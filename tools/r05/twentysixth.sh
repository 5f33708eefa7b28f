#!/bin/bash
# round 5, call 26: kernel shares of the reference's own workloads (fingerprint: 680,000 x 2048 counts, int8 form; morgan: 800,000 x 1024 bits, fp4 form)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05/fp_prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fp -- python3 $R/bench.py --workload fingerprint --steps 2 --warmup 1 --no-cpu-baseline > $O/fp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mg -- python3 $R/bench.py --workload morgan --n-corpus 800000 --steps 2 --warmup 1 --no-cpu-baseline > $O/mg.log 2>&1
for d in fp mg; do f=$(ls $O/$d/*/*kernel_stats.csv | head -1); echo "== $d"; head -9 $f | cut -c1-150; done
