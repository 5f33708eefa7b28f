#!/bin/bash
L=tools/scan_lab
out=gpurun_out/lab4.log
: > $out
for v in 5 6 7 8; do $L $v 4 0 500 1 >> $out 2>&1; done
for v in 2 5 8 4 6 7; do
  for f in 0 5; do timeout 120 $L $v 4 $f 3907 3 >> $out 2>&1; done
done
for s in 2 3 5 6; do timeout 120 $L 6 $s 0 3907 3 >> $out 2>&1; done
