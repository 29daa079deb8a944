#!/bin/bash
# round 4, first GPU call: bisect of the wavefront regression, record-major store model, placement per box
O=gpurun_out/r04_call1; mkdir -p $O
prev=""
for t in b0be7b5 57dcdcb 7dc0e0c; do
  timeout 300 python profiles/micro/wavefront_bisect.py variants/$t $t $prev > $O/bisect_$t.json 2> $O/bisect_$t.err
  prev="$prev gpurun_out/wf_$t.npy"
done
timeout 300 python profiles/micro/wavefront_bisect.py . head $prev > $O/bisect_head.json 2> $O/bisect_head.err
MPLX_DONE_FLAG=0 timeout 300 python profiles/micro/wavefront_bisect.py . head_nodone $prev > $O/bisect_head_nodone.json 2> $O/bisect_head_nodone.err
MPLX_SERVICE=0 timeout 300 python profiles/micro/wavefront_bisect.py . head_nosvc $prev > $O/bisect_head_nosvc.json 2> $O/bisect_head_nosvc.err
timeout 120 profiles/micro/write_layout 6 > $O/write_layout.txt 2>&1
for i in 1 2 3; do timeout 200 python profiles/micro/placement_table.py 8 >> $O/placement.txt 2>> $O/placement.err; done
tail -n 3 $O/*.json $O/write_layout.txt $O/placement.txt
