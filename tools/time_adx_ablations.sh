#!/bin/bash
# adx_encode_fs18_direct_kernel at configs[2]: the product and timing-only builds (tools/build_variants.sh with
# VARIED=adx_kernels and -DVGA_ADX_ABLATE=1 no crumbs, 2 no stores, 4 no pre-scan, 10 no stores and no loads after the first)
for v in "" xa1 xa2 xa4 xa10; do
  if [ -n "$v" ]; then [ -f $GRAFT_REPO_ROOT/tools/variants/libvga_$v.so ] || continue; export VGAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/variants/libvga_$v.so; fi
  echo "== ${v:-product}"
  CALLS=4 bash $(dirname $0)/prof_kernels.sh python tools/adx_encode_once.py 2>&1 | grep -E "direct|fixup"
done
