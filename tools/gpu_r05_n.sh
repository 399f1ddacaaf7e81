#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python tools/time_encode_variants.py > $O/encode_variants.log 2>&1
grep -v amdgpu.ids $O/encode_variants.log | cut -c1-200 | tail -3
echo "== small batches, product (layout 4 ms, layout 8 ms)"; timeout 600 python tools/time_layouts_small.py 2>&1 | grep -v amdgpu | tee $O/layouts_small_product.log
echo "== small batches, exact passes"; VGAUDIO_HIP_LIBRARY=tools/variants/libvga_exactpasses.so timeout 600 python tools/time_layouts_small.py 2>&1 | grep -v amdgpu | tee $O/layouts_small_exactpasses.log
for t in test_gpu_gcadpcm test_gpu_ragged test_gpu_signal_classes test_gpu_full_size test_gpu_shards test_gpu_golden test_gpu_host_pipeline test_gpu_dsp; do
  timeout 900 python -m pytest tests/$t.py -q -m gpu > $O/pytest_$t.log 2>&1
  echo "== $t: $(grep -v amdgpu.ids $O/pytest_$t.log | tail -1)"
  grep -E "^(FAILED|ERROR)" $O/pytest_$t.log | cut -c1-250 | head -20
done
