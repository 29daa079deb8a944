#!/bin/bash
O=gpurun_out/r04_call4; mkdir -p $O
for w in none main e2e edges main+e2e+edges; do timeout 300 python profiles/micro/wavefront_leg3.py $w >> $O/leg3.txt 2>> $O/leg3.err; done
cat $O/leg3.txt; tail -n 5 $O/leg3.err
