#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_encode_variants.py > $O/encode_variants.log 2>&1
grep -v amdgpu.ids $O/encode_variants.log | cut -c1-200 | tail -3
echo "== mid batches, product"; timeout 600 python tools/time_layouts_mid.py 2>&1 | grep -v amdgpu | tee $O/mid_product.log
echo "== mid batches, exact passes"; VGAUDIO_HIP_LIBRARY=tools/variants/libvga_exactpasses.so timeout 600 python tools/time_layouts_mid.py 2>&1 | grep -v amdgpu | tee $O/mid_exactpasses.log
echo "== mid batches, product again"; timeout 600 python tools/time_layouts_mid.py 2>&1 | grep -v amdgpu | tee $O/mid_product2.log
