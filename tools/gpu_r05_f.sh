#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_encode_variants.py > $O/encode_variants.log 2>&1
grep -v amdgpu.ids $O/encode_variants.log | cut -c1-200 | tail -5
for t in test_gpu_gcadpcm test_gpu_adx test_gpu_ragged test_gpu_signal_classes test_gpu_full_size; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu > $O/pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/pytest_$t.log | cut -c1-250 | head -20
done
for cls in clipped_square sine440; do
  timeout 300 python tools/time_signal_class.py $cls 2>&1 | grep " ms"
done
timeout 600 python bench.py > $O/bench_default.json.log 2> $O/bench_default.err
tail -c 600 $O/bench_default.json.log; tail -3 $O/bench_default.err
