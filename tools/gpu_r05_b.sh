#!/bin/bash
# round 5, second visit: ragged + signal-class parity, the default bench line with signal_sensitivity, HCA encode stage times
O=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_signal_classes.py -x -q -m gpu > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/pytest.log | tail -15 | cut -c1-300
timeout 600 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err
tail -c 6000 $O/bench_default.json.log; tail -3 $O/bench_default.err
timeout 600 python tools/time_hca_decode.py > $O/hca_encode_stages.log 2>&1
grep -v amdgpu.ids $O/hca_encode_stages.log | cut -c1-200 | tail -10
