#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_encode_variants.py > $O/encode_variants.log 2>&1
grep -v amdgpu.ids $O/encode_variants.log | cut -c1-200 | tail -3
VGAUDIO_HIP_LIBRARY=vgaudio_amd/libvgaudio_hip_stats.so timeout 300 python tools/signal_cold_rates.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1
grep -v amdgpu.ids $O/pytest_gpu.log | tail -4 | cut -c1-300
grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | cut -c1-250 | head -20
