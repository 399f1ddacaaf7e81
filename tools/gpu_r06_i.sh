#!/bin/bash
# round 6, call i: the host pipeline's rows by transfer kernels: parity (host pipeline, ragged tests), then the ragged calls
# of the three codecs with transfer kernels and with one copy per row, same box
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in test_gpu_host_pipeline test_gpu_ragged; do
  timeout 1200 python -m pytest tests/$t.py -q -m gpu -x > $O/r06_i_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_i_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_i_pytest_$t.log | cut -c1-250 | head -20
done
timeout 1500 python tools/time_ragged_host.py --codecs gc adx hca --orders 0 --transfer 1 0 > $O/r06_i_ragged_host_transfer.log 2>&1
grep -v amdgpu.ids $O/r06_i_ragged_host_transfer.log | cut -c1-260
