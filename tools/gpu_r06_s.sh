#!/bin/bash
# round 6, call s: gc_coefs_kernel: priority maps, wave end times
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 800 python tools/time_coefs_prio.py tools/variants/libvga_coef*.so 2>&1 | grep -v amdgpu.ids | tee $O/r06_s_coefs_priority_maps2.log
for v in tools/variants/libvga_ts*.so; do
  echo "== $v" | tee -a $O/r06_s_coefs_wave_ends2.log
  VGAUDIO_HIP_LIBRARY=$v timeout 300 python tools/time_wave_ends.py 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tee -a $O/r06_s_coefs_wave_ends2.log
done
