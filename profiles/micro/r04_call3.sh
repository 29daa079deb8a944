#!/bin/bash
O=gpurun_out/r04_call3; mkdir -p $O
timeout 300 python profiles/micro/wavefront_leg.py > $O/leg.json 2> $O/leg.err
MPLX_DONE_FLAG=0 timeout 300 python profiles/micro/wavefront_leg.py > $O/leg_nodone.json 2> $O/leg_nodone.err
cat $O/leg.json $O/leg_nodone.json; tail -3 $O/*.err
