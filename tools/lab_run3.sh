#!/bin/bash
L=tools/scan_lab
out=gpurun_out/lab3.log
: > $out
for f in 9 8 1 0 13 5; do
  for s in 4; do timeout 120 $L 4 $s $f 3907 3 >> $out 2>&1; done
done
