for v in "" xa1 xa2 xa4 xa10 xa14; do
  if [ -n "$v" ]; then export VGAUDIO_HIP_LIBRARY=$GRAFT_REPO_ROOT/tools/variants/libvga_$v.so; fi
  echo "== variant ${v:-product}"
  CALLS=4 bash tools/prof_kernels.sh python tools/adx_encode_once.py 2>&1 | grep -E "direct"
done
