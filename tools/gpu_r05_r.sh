#!/bin/bash
# round 5: ADX length buckets with the channels' own frame counts -- the ragged and ADX tests, then both bucket orders again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_adx.py -m gpu -x -q --timeout=600 > $O/pytest_ragged_adx.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest_ragged_adx.log
VGA_HIP_PIPELINE_TIMELINE=1 timeout 600 python tools/time_ragged_host.py --codecs adx --reps 2 > $O/ragged_host_orders_own_frames.log 2>&1; echo "ragged host rc=$?"
grep -v amdgpu $O/ragged_host_orders_own_frames.log | grep -v "^timeline" | tail -8
grep "^timeline" $O/ragged_host_orders_own_frames.log | tail -22
