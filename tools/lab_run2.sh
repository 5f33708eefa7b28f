#!/bin/bash
L=tools/scan_lab
out=gpurun_out/lab2.log
: > $out
for v in 2 3 4; do $L $v 1 0 500 2 >> $out 2>&1; $L $v 4 0 500 1 >> $out 2>&1; done
for v in 1 2 3 4; do
  for s in 1 4; do
    for f in 0 1 2; do
      timeout 120 $L $v $s $f 3907 3 >> $out 2>&1
    done
  done
done
