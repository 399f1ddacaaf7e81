#!/bin/bash
# round 5: uploads on two or three streams (a feeder thread each) instead of one -- the equal-length GC call and the ragged ADX call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/sweep_host_pipeline.py --shapes "0,0,0,0;-2,0,0,0;-3,0,0,0;2,0,0,0;0,0,0,0" > $O/sweep_feeder_streams.log 2>&1; echo "sweep rc=$?"
grep -v amdgpu $O/sweep_feeder_streams.log | cut -c1-420
timeout 600 python tools/time_ragged_host.py --codecs adx --orders 1 --feeders 0 -2 -3 > $O/ragged_host_feeder_streams.log 2>&1; echo "ragged rc=$?"
grep -v amdgpu $O/ragged_host_feeder_streams.log | tail -8
