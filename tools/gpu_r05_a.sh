#!/bin/bash
# round 5, first visit: the advisor's fixes on hardware, the co-run experiment, two encoder ablations
O=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ragged.py -x -q -m gpu -k "longer_than or empty_channels or adx_ragged or persistent_workgroups" > $O/pytest_ragged.log 2>&1
tail -3 $O/pytest_ragged.log
timeout 400 python tools/time_corun.py > $O/corun.log 2>&1
grep -v amdgpu.ids $O/corun.log | tail -12
timeout 600 python tools/time_encode_variants.py > $O/encode_variants.log 2>&1
grep -v amdgpu.ids $O/encode_variants.log | cut -c1-200 | tail -6
