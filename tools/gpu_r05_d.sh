#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python tools/debug_edge_encode.py > $O/debug_edge.log 2>&1
grep -v amdgpu.ids $O/debug_edge.log | tail -40 | cut -c1-300
for t in test_gpu_gcadpcm test_gpu_golden test_gpu_hca test_gpu_host_pipeline test_gpu_ragged test_gpu_shards test_gpu_signal_classes test_gpu_wave; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu > $O/pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/pytest_$t.log | cut -c1-250 | head -20
done
