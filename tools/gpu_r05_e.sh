#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $O
cd $GRAFT_REPO_ROOT
for t in test_gpu_gcadpcm test_gpu_signal_classes test_gpu_full_size; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu > $O/pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/pytest_$t.log | cut -c1-250 | head -20
done
cd /tmp && export TMPDIR=/tmp
for cls in clipped_square sine440; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$cls -o $cls -- python $GRAFT_REPO_ROOT/tools/time_signal_class.py $cls > $O/signal_$cls.log 2>&1
  grep -v amdgpu.ids $O/signal_$cls.log | grep " ms" 
  f=$(find $O/prof_$cls -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cut -d, -f1-6 "$f" | head -16
done
