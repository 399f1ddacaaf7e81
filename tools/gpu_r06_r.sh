#!/bin/bash
# round 6, call r: gc_coefs_kernel with the scan partition (+ issue-priority variants): parity of the product build, times
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gcadpcm.py -q -m gpu -x -k "coef" > $O/r06_r_pytest_coefs.log 2>&1
echo "== coefs tests: $(grep -v amdgpu.ids $O/r06_r_pytest_coefs.log | tail -1)"
grep -E "^(FAILED|ERROR)" $O/r06_r_pytest_coefs.log | cut -c1-250 | head
timeout 800 python tools/time_coefs_prio.py tools/variants/libvga_scan*.so 2>&1 | grep -v amdgpu.ids | tee $O/r06_r_coefs_scan_partition.log
