#!/bin/bash
# round 6, call g: the ADX encoder's REPAIR launch: parity (ADX tests, signal classes), then the tone / square / synthetic timings
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in test_gpu_adx test_gpu_signal_classes test_gpu_ragged; do
  timeout 1200 python -m pytest tests/$t.py -q -m gpu -x > $O/r06_g_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_g_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_g_pytest_$t.log | cut -c1-250 | head -20
done
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -m gpu -k "adx" > $O/r06_g_pytest_full_adx.log 2>&1
echo "== full size adx: $(grep -v amdgpu.ids $O/r06_g_pytest_full_adx.log | tail -1)"
for c in sine440 clipped_square synthetic; do timeout 300 python tools/time_signal_class.py $c 2>&1 | grep -v amdgpu.ids; done | tee $O/r06_g_signal_class_times.log
