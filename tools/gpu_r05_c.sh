#!/bin/bash
# round 5, third visit: the whole GPU suite on the new encoder cold block / HCA cost look-up / decoder repair launches, the bench
# line, HCA encode stage times
O=$GRAFT_REPO_ROOT/gpurun_out/r05c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1
grep -v amdgpu.ids $O/pytest_gpu.log | tail -25 | cut -c1-400
timeout 600 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json.log; tail -3 $O/bench_default.err
timeout 600 python tools/time_hca_decode.py > $O/hca_encode_stages.log 2>&1
grep -v amdgpu.ids $O/hca_encode_stages.log | cut -c1-200 | tail -10
