#!/bin/bash
# round 5: is the two-stream upload of the ragged calls steady?  Every call's time, one stream (feeders 1) against the product (0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python tools/time_ragged_host.py --orders 1 --feeders 1 0 --reps 6 > $O/ragged_host_steady.log 2>&1; echo "rc=$?"
grep -v amdgpu $O/ragged_host_steady.log | tail -10 | cut -c1-400
