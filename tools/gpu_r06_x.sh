#!/bin/bash
# round 6, call x: where the coefficient kernel's waves end -- by XCD, by CU, by arrival order; mixed channels and identical rows
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
for sig in "" slow_channel_93 white_full_scale; do
  echo "== signal '${sig:-synthetic}'" | tee -a $O/r06_x_coefs_wave_ends_where.log
  VGAUDIO_HIP_LIBRARY=tools/variants/libvga_tsP.so timeout 300 python tools/time_wave_ends.py ${sig:+--signal $sig} 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee -a $O/r06_x_coefs_wave_ends_where.log
done
