#!/bin/bash
# round 6, call e: GC encoder cold block trimmed: parity (gcadpcm + signal classes + host emulator is CPU) and timing
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in ; do
  timeout 1200 python -m pytest tests/$t.py -q -m gpu -x > $O/r06_e_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_e_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_e_pytest_$t.log | cut -c1-250 | head -20
done
timeout 600 python tools/time_encode_variants.py > $O/r06_e_encode_variants.log 2>&1
grep -v amdgpu.ids $O/r06_e_encode_variants.log | cut -c1-200
