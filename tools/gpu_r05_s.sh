#!/bin/bash
# round 5: the ragged ADX / HCA host calls, both bucket orders, chunks of at most 1024 / 256 / 128 units
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python tools/time_ragged_host.py --chunk-units 0 256 128 > $O/ragged_host_orders_chunks.log 2>&1; echo "ragged host rc=$?"
grep -v amdgpu $O/ragged_host_orders_chunks.log | tail -30
