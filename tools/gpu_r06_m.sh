#!/bin/bash
# round 6, call m: the seam and chain launches in the lane-per-candidate layout (batches below 512 channels): parity, channel scaling
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in test_gpu_gcadpcm test_gpu_golden; do
  timeout 1200 python -m pytest tests/$t.py -q -m gpu -x > $O/r06_m_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_m_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_m_pytest_$t.log | cut -c1-250 | head -20
done
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -m gpu -k "segment or fallback" > $O/r06_m_pytest_full.log 2>&1
echo "== full size (segments): $(grep -v amdgpu.ids $O/r06_m_pytest_full.log | tail -1)"
timeout 900 python tools/time_encode_channels.py 1 8 64 96 128 256 384 512 1024 > $O/r06_m_channel_scaling.log 2>&1
grep -v amdgpu.ids $O/r06_m_channel_scaling.log
