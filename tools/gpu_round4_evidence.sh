#!/bin/bash
# Round 4's experiment logs in one visit (needs the libraries tools/build_variants.sh made: ts, nothird, ab*, k*):
# what every profiles/r04_a_* / r04_b_* file is copied from.
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
cd $GRAFT_REPO_ROOT
{ echo "# plain grid, four pieces per channel (vga_testing_gc_encoder_persistent_this_thread(1))";
  VGAUDIO_HIP_LIBRARY=tools/variants/libvga_ts.so timeout 300 python tools/time_wave_ends.py --persistent 1;
  echo "# persistent workgroups, the launcher's schedule";
  VGAUDIO_HIP_LIBRARY=tools/variants/libvga_ts.so timeout 300 python tools/time_wave_ends.py --persistent 0; } > $O/a_wave_ends.log 2>&1
timeout 600 python tools/time_encode_persistent.py --channels 4096 2048 1024 512 --pieces 0 8 12 16 24 32 64 > $O/a_encode_persistent.log 2>&1
timeout 600 python tools/time_encode_schedule.py --channels 4096 3072 2048 1024 8192 > $O/a_encode_schedules.log 2>&1
timeout 600 python tools/time_ragged_schedule.py > $O/a_ragged_schedules.log 2>&1
mkdir -p $O/hold && mv tools/variants/libvga_*.so $O/hold/
for set in "ab1 ab2 ab7 ab32 ab56 k1:b_coefs_ablations" "k1 k2 k3 k6:b_coefs_chunks" "nothird:b_third_trip_ablation"; do
  names=${set%%:*}; out=${set##*:}
  for n in $names; do cp $O/hold/libvga_$n.so tools/variants/; done
  timeout 600 python tools/time_encode_variants.py > $O/$out.log 2>&1
  rm -f tools/variants/libvga_*.so
done
mv $O/hold/libvga_*.so tools/variants/; rmdir $O/hold
[ -x tools/variants/ubench_f64 ] && tools/variants/ubench_f64 > $O/b_ubench_f64.log 2>&1
for f in a_wave_ends a_encode_persistent a_encode_schedules a_ragged_schedules b_coefs_ablations b_coefs_chunks b_third_trip_ablation; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | cut -c1-260 | tail -12; done
