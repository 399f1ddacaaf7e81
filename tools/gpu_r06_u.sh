#!/bin/bash
# round 6, call u: gc_coefs_kernel with scan partition + priorities by pass: every GC test (uniform, ragged, shards, full size)
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for t in test_gpu_gcadpcm test_gpu_ragged test_gpu_shards test_gpu_full_size test_gpu_host_batch; do
  [ -f tests/$t.py ] || continue
  timeout 1500 python -m pytest tests/$t.py -q -m gpu -x > $O/r06_u_pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/r06_u_pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/r06_u_pytest_$t.log | cut -c1-250 | head
done
timeout 300 python tools/time_coefs_variants.py --channels 4096 8192 3072 1024 128 1 2>&1 | grep -v amdgpu.ids | tee $O/r06_u_coefs_variants.log
