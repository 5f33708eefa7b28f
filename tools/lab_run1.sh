#!/bin/bash
# first laboratory sweep: variants x splits x ablation flags
L=tools/scan_lab
out=gpurun_out/lab1.log
: > $out
$L 1 1 0 500 2 >> $out 2>&1     # correctness check of variant 1 vs 0
$L 1 8 0 496 2 >> $out 2>&1
for v in 0 1; do
  for s in 1 4 8; do
    for f in 0 1 2; do
      timeout 120 $L $v $s $f 3907 3 >> $out 2>&1
    done
  done
done
