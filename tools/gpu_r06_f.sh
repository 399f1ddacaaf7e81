#!/bin/bash
# round 6, call f: padded ADX streams through the time-piece encoder: parity and timing
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_adx.py -q -m gpu -x > $O/r06_f_pytest_adx.log 2>&1
grep -v amdgpu.ids $O/r06_f_pytest_adx.log | tail -8 | cut -c1-300
timeout 600 python tools/time_adx_looping.py > $O/r06_f_adx_looping.log 2>&1
grep -v amdgpu.ids $O/r06_f_adx_looping.log | tail -5
