#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
VGA_HIP_PIPELINE_TIMELINE=1 timeout 900 python tools/time_ragged_host.py --codecs gc --orders 0 --transfer 1 0 --reps 1 > $O/r06_j_timeline.log 2>&1
grep -v amdgpu.ids $O/r06_j_timeline.log | cut -c1-220 | tail -60
