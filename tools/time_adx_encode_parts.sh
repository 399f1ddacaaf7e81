#!/bin/bash
# The ADX encoder's kernels at configs[2] for several piece counts and numbers of fix-up waves (rocprofv3 kernel trace of
# tools/adx_encode_once.py; the pieces through the test hook, the waves through VGA_HIP_ADX_FIXUP_WAVES)
for seg in ${SEGS:-16 32}; do for w in ${WAVES:-512 1024 2048}; do
  echo "== pieces $seg, fix-up waves $w"
  VGA_ADX_SEGMENTS=$seg VGA_HIP_ADX_FIXUP_WAVES=$w CALLS=4 bash $(dirname $0)/prof_kernels.sh python tools/adx_encode_once.py 2>&1 | grep -E "direct|fixup"
done; done
