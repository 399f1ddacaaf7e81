#!/bin/bash
# round 5: the ragged ADX / HCA host calls with their length buckets shortest first (until now) and longest first, one box;
# the ragged tests on the new order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ragged.py -m gpu -x -q --timeout=600 > $O/pytest_ragged.log 2>&1; echo "ragged tests rc=$?"; tail -3 $O/pytest_ragged.log
timeout 600 python tools/time_ragged_host.py > $O/ragged_host_orders.log 2>&1; echo "ragged host rc=$?"; grep -v amdgpu $O/ragged_host_orders.log | tail -12
